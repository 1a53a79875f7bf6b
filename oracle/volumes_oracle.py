"""Oracle (test infrastructure only) for inference/tiles_3d.py of the reference (numpy restatement of the parts that
work there).  Citations are ``pytorch_toolbelt/inference/tiles_3d.py:LINE``."""
import numpy as np


def _triple(v):
    if isinstance(v, (tuple, list, np.ndarray)):
        if len(v) != 3:
            raise ValueError
        return [int(x) for x in v]
    return [int(v)] * 3


def slicer_geometry(volume_shape, voxel_size, voxel_step):
    """tiles_3d.py:14-107.  Per axis: n = max(1, ceil((S - overlap) / step)); extra = step*n - (S - overlap);
    pad_before = extra // 2; tile origins 0, step, ... while origin + size <= S + extra.  Tiles are enumerated depth
    outermost, columns innermost.  Returns dict(size, step, pad_before, pad_after, target_shape, starts[N,3], bbox_starts[N,3])."""
    shape = [int(s) for s in volume_shape[:3]]
    size, step = _triple(voxel_size), _triple(voxel_step)
    for a in range(3):
        if step[a] < 1 or step[a] > size[a]:
            raise ValueError                                                            # :48-53
    before, after, target, axes = [], [], [], []
    for a in range(3):
        overlap = size[a] - step[a]                                                     # :55
        n = max(1, -(-(shape[a] - overlap) // step[a]))                                 # :59 ceil
        extra = step[a] * n - (shape[a] - overlap)                                      # :60
        before.append(extra // 2)                                                       # :61
        after.append(extra - extra // 2)                                                # :62
        target.append(shape[a] + extra)                                                 # :135-137
        axes.append(list(range(0, shape[a] + extra - size[a] + 1, step[a])))            # :80-82
    starts = [(i, j, k) for i in axes[0] for j in axes[1] for k in axes[2]]             # :84-86
    starts = np.asarray(starts, dtype=np.int64).reshape(-1, 3)
    return dict(size=size, step=step, pad_before=before, pad_after=after, target_shape=tuple(target), starts=starts,
                bbox_starts=starts - np.asarray(before, dtype=np.int64))                # :92-96


def split(volume, geom, value=0):
    """tiles_3d.py:109-121: constant-pad, then copy every roi."""
    pad = list(zip(geom["pad_before"], geom["pad_after"])) + [(0, 0)] * (volume.ndim - 3)
    padded = np.pad(volume, pad, mode="constant", constant_values=value)
    d, h, w = geom["size"]
    return [padded[z:z + d, y:y + h, x:x + w].copy() for z, y, x in geom["starts"]]


def crop_to_original(volume, geom, volume_shape):
    """tiles_3d.py:162-163 with the roi of :63-67."""
    b = geom["pad_before"]
    return volume[b[0]:b[0] + volume_shape[0], b[1]:b[1] + volume_shape[1], b[2]:b[2] + volume_shape[2]]


def merger_new(volume_shape, channels, weight, dtype=np.float32):
    """VolumeMerger.__init__, tiles_3d.py:174-185."""
    shape = tuple(int(s) for s in volume_shape)
    return dict(weight=np.asarray(weight)[None].astype(dtype), volume=np.zeros((channels,) + shape, dtype=dtype),
                norm_mask=np.zeros((1,) + shape, dtype=dtype))


def merger_integrate(state, batch, starts):
    """VolumeMerger.integrate_batch, tiles_3d.py:195-208: sequential ``volume[:, roi] += tile * weight; norm[:, roi] += weight``."""
    if len(batch) != len(starts):
        raise ValueError("Number of images in batch does not correspond to number of coordinates")
    vol, nrm, w = state["volume"], state["norm_mask"], state["weight"]
    d, h, ww = w.shape[1:]
    for tile, (z, y, x) in zip(np.asarray(batch), starts):
        z, y, x = int(z), int(y), int(x)
        vol[:, z:z + d, y:y + h, x:x + ww] += (tile * w).astype(vol.dtype)             # :205-206
        nrm[:, z:z + d, y:y + h, x:x + ww] += w                                        # :207
    return state


def merger_merge(state):
    """VolumeMerger.merge, tiles_3d.py:210-211: plain division."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return state["volume"] / state["norm_mask"]
