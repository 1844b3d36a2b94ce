from . import kmeans, knn

__all__ = ["kmeans", "knn"]
