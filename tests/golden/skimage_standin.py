"""Stand-in for `skimage.feature.match_descriptors` used by the golden generators (build container only).

scikit-image is a third-party dependency of the reference (requirements.txt:13, unpinned), absent from the reference
tree and from this image, and there is no network to install it.  What the reference calls (utils/matcher.py:227-230) is
restated here from scikit-image's documented behaviour ON TOP OF THE REAL DEPENDENCY that does the arithmetic,
`scipy.spatial.distance.cdist` (present: the version is recorded in every fixture this produces):

    distances = cdist(descriptors1, descriptors2, metric)           # float64
    indices2  = argmin(distances, axis=1)                           # first index on ties
    cross_check: keep i where argmin(distances, axis=0)[indices2[i]] == i
    max_distance < inf: keep pairs with distances[i, j] < max_distance (strict)
    max_ratio < 1: Lowe's ratio test on the two smallest distances of each row (unused by the reference)
    -> column_stack((indices1, indices2)), ascending in indices1

The oracle (oracle/kpb_oracle.c) is a separate restatement in C with its own loops; fixtures made with THIS function pin it
(and through it the HIP kernels) on scipy's arithmetic rather than on itself.  Still "parity unpinned" for the few
lines of glue, which only the real scikit-image could pin."""
import numpy as np
import scipy
from scipy.spatial.distance import cdist

SCIPY_VERSION = scipy.__version__


def match_descriptors(descriptors1, descriptors2, metric=None, p=2, max_distance=np.inf, cross_check=True, max_ratio=1.0):
    if descriptors1.shape[1] != descriptors2.shape[1]:
        raise ValueError("Descriptor length must equal.")
    if metric is None:
        metric = "hamming" if np.issubdtype(descriptors1.dtype, np.bool_) else "euclidean"
    kwargs = {"p": p} if metric == "minkowski" else {}
    distances = cdist(descriptors1, descriptors2, metric=metric, **kwargs)
    indices1 = np.arange(descriptors1.shape[0])
    indices2 = np.argmin(distances, axis=1)
    if cross_check:
        matches1 = np.argmin(distances, axis=0)
        mask = indices1 == matches1[indices2]
        indices1, indices2 = indices1[mask], indices2[mask]
    if max_distance < np.inf:
        mask = distances[indices1, indices2] < max_distance
        indices1, indices2 = indices1[mask], indices2[mask]
    if max_ratio < 1.0:
        best = distances[indices1, indices2]
        distances[indices1, indices2] = np.inf
        second = distances[indices1, np.argmin(distances[indices1], axis=1)]
        second[second == 0] = np.finfo(np.double).eps
        mask = best / second < max_ratio
        indices1, indices2 = indices1[mask], indices2[mask]
    return np.column_stack((indices1, indices2))


def distances_of(descriptors1, descriptors2, pairs):
    """float64 cdist values of the returned pairs (what the fixtures store next to them)."""
    d = cdist(descriptors1, descriptors2, metric="euclidean")
    return d[pairs[:, 0], pairs[:, 1]]
