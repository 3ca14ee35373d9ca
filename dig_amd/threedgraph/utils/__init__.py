from .geometric_computing import xyz_to_dat

__all__ = ['xyz_to_dat']
