"""Product front end (host C++ in libpwicp.so): supervoxel labels for bench.py / demos."""
from .binding import frontend_segment


def segment(cloud, sv_resolution, knn=45):
    return frontend_segment(cloud, sv_resolution, knn)
