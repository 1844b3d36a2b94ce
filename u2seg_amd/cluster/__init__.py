from . import kmeans

__all__ = ["kmeans"]
