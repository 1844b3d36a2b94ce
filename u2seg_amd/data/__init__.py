from .synthetic import make_synthetic_batch, synthetic_sample

__all__ = ["make_synthetic_batch", "synthetic_sample"]
