"""Mean-absolute-error evaluator; API of dig/threedgraph/evaluation/eval.py:12-34."""
import numpy as np
import torch


class ThreeDEvaluator:
    r"""Evaluator for the 3D datasets (QM9, MD17).  ``eval({'y_true': .., 'y_pred': ..}) -> {'mae': float}``;
    both entries must be numpy arrays or both torch tensors, of identical shape."""

    def eval(self, input_dict):
        assert 'y_pred' in input_dict
        assert 'y_true' in input_dict
        y_pred, y_true = input_dict['y_pred'], input_dict['y_true']
        both_np = isinstance(y_true, np.ndarray) and isinstance(y_pred, np.ndarray)
        both_t = isinstance(y_true, torch.Tensor) and isinstance(y_pred, torch.Tensor)
        assert both_np or both_t
        assert y_true.shape == y_pred.shape
        if both_t:
            return {'mae': (y_pred - y_true).abs().mean().cpu().item()}
        return {'mae': float(np.abs(y_pred - y_true).mean())}
