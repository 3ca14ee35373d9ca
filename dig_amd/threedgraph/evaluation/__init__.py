from .eval import ThreeDEvaluator

__all__ = ['ThreeDEvaluator']
