"""Host-side mirror of the reference's voxeliser (datasets/MVSEC/utils.py:215-281, called from
datasets/MVSEC/mvsec_dataset.py:179-180): events -> per-pixel two-polarity count frames, on the MI355X.

The MVSEC loaders themselves (HDF5 reading, rectification maps, split indices) are out of scope (SURVEY.md §2 rows 10-12:
the dataset download, h5py, cv2, skimage are absent); this is the one compute step between them and the network."""
import torch

from . import _lib

LIDAR_FPS = 20
H, W = 260, 346


def frame_bounds(n_chunks: int, num_frames_per_depth_map: int):
    """Open-interval bounds of every frame, in python float64 arithmetic in the reference's expression order
    (utils.py:259-260) so that an event sitting exactly on a boundary is treated identically."""
    fps = num_frames_per_depth_map * LIDAR_FPS
    start, end = [], []
    for numchunk in range(n_chunks):
        for numframe in range(num_frames_per_depth_map):
            start.append(numchunk * num_frames_per_depth_map * 1 / fps + numframe * 1 / fps)
            end.append(numchunk * num_frames_per_depth_map * 1 / fps + numframe * 1 / fps + 1 / fps)
    return torch.tensor(start, dtype=torch.float64), torch.tensor(end, dtype=torch.float64)


def mvsecCumulateSpikesIntoFrames(events: torch.Tensor, n_chunks: int, num_frames_per_depth_map: int = 1) -> torch.Tensor:
    """events: [E, 4] float64 HIP tensor (X, Y, TIME, POLARITY), time-sorted as the reference keeps them (the first row's
    time is the offset that is removed).  Returns [n_chunks, num_frames_per_depth_map, 2, 260, 346] float32 counts —
    what the reference's function returns as its first result (as float64) and train.py:194-197 casts to float32."""
    assert num_frames_per_depth_map in [1, 2, 5, 10, 25], 'num_frames_per_depth_map must divide 50 ! Choose another ' \
                                                          'value among [1, 2, 5, 10, 25] ...'
    events = events.contiguous()
    start, end = (t.to(events.device) for t in frame_bounds(n_chunks, num_frames_per_depth_map))
    G = n_chunks * num_frames_per_depth_map
    counts = torch.empty((G, 2, H, W), dtype=torch.int32, device=events.device)
    _lib.voxelize(events, start, end, counts, H, W)
    return counts.to(torch.float32).view(n_chunks, num_frames_per_depth_map, 2, H, W)
