"""Drop-in ``simple_knn`` for WildGaussians on MI355X: ``from simple_knn._C import distCUDA2`` (wildgaussians/method.py:25)."""
