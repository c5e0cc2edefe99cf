"""TEST-ONLY stand-in for `plyfile` (absent from this image): method.py imports the two names at module level
(method.py:23) and uses them only in save_ply (method.py:1212-1247), which the harness never calls."""


class PlyElement:
    @staticmethod
    def describe(*a, **k):
        raise NotImplementedError("plyfile stand-in: save_ply is outside the tested path")


class PlyData:
    def __init__(self, *a, **k):
        raise NotImplementedError("plyfile stand-in: save_ply is outside the tested path")
