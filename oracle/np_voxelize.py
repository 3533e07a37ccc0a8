"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's voxeliser
`mvsecCumulateSpikesIntoFrames` (/root/reference/datasets/MVSEC/utils.py:215-281): events [E, 4] = (X, Y, TIME, POLARITY)
-> per-pixel 2-polarity count frames, `num_frames_per_depth_map` frames per 50 ms label interval.

Follows the reference line by line where results depend on it:
  * the temporal offset is removed first (utils.py:248-250: t -= events[0, 2]);
  * frame (numchunk, numframe) covers the OPEN interval (start_ts, end_ts) with
        start_ts = numchunk*nfpdm*1/fps + numframe*1/fps,  end_ts = start_ts-expression + 1/fps      (utils.py:259-261)
    evaluated in float64 in exactly that order (an event on a boundary is dropped, an event between start[g+1] and an
    end[g] that lies one ulp above it is counted in BOTH frames — kept);
  * channel 0 counts POLARITY == 1, channel 1 everything else; pixel = (int(Y), int(X)) truncating toward zero (:268-275).
Deviation (documented reference defect): the reference keeps rectified coordinates equal to 346 / 260
(utils.py:52-55) and would then raise IndexError here; such events are skipped.
tests/golden/make_golden.py pins this file to the reference's own function (extracted from utils.py in the build container)."""
import numpy as np

LIDAR_FPS = 20
H, W = 260, 346


def frame_bounds(n_chunks, nfpdm):
    """start/end tables [n_chunks * nfpdm] in float64, evaluated exactly like utils.py:259-260."""
    fps = nfpdm * LIDAR_FPS
    start = np.empty(n_chunks * nfpdm, np.float64)
    end = np.empty(n_chunks * nfpdm, np.float64)
    for numchunk in range(n_chunks):
        for numframe in range(nfpdm):
            start[numchunk * nfpdm + numframe] = numchunk * nfpdm * 1 / fps + numframe * 1 / fps
            end[numchunk * nfpdm + numframe] = numchunk * nfpdm * 1 / fps + numframe * 1 / fps + 1 / fps
    return start, end


def cumulate_spikes_into_frames(events, n_chunks, nfpdm=1):
    """events [E, 4] float64 (not modified) -> frames [n_chunks, nfpdm, 2, 260, 346] float64 counts."""
    ev = np.asarray(events, np.float64)
    t = ev[:, 2] - ev[0, 2]
    start, end = frame_bounds(n_chunks, nfpdm)
    frames = np.zeros((n_chunks * nfpdm, 2, H, W), np.float64)
    x = ev[:, 0].astype(np.int64)          # int(): truncation toward zero
    y = ev[:, 1].astype(np.int64)
    ch = np.where(ev[:, 3] == 1, 0, 1)
    inside = (x >= 0) & (x < W) & (y >= 0) & (y < H)
    for g in range(n_chunks * nfpdm):
        m = (t > start[g]) & (t < end[g]) & inside
        np.add.at(frames[g], (ch[m], y[m], x[m]), 1.0)
    return frames.reshape(n_chunks, nfpdm, 2, H, W)
