from . import kmeans, knn, select

__all__ = ["kmeans", "knn", "select"]
