"""GEMM algorithm selection for the library GEMMs of the decoder (hipBLASLt / rocBLAS through PyTorch's TunableOp).

The fp32 projection / data-gradient / weight-gradient GEMMs of the decoder are skinny (K = 32..512 against M up to 1.8 M rows); the
default heuristic picks poorly for several of them.  PyTorch-ROCm's TunableOp times every hipBLASLt / rocBLAS solution once per shape and
records the winner; the record for this network's shapes on gfx950 (measured on an MI355X, `tools/tune_gemms.sh`) is tracked in
`stereospike_amd/tunableop/tunableop_results.csv` and loaded READ-ONLY here — the GEMM counterpart of the MIOpen find-db seed
(`miopen_cache.py`).  Shapes that are not in the record fall back to the library default; a record made with other library versions is
ignored by TunableOp's validators.  Measured: 58.8 -> 55.6 ms per training step (profiles/README.md)."""
import os
import shutil

SEED = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tunableop', 'tunableop_results.csv')
WORK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'tunableop')


def enable(device_index: int = 0, tuning: bool = False, filename: str = None):
    """Call after `import torch`, before the first GEMM.  tuning=True times unseen shapes and appends them to `filename`."""
    import torch
    if not (torch.cuda.is_available() and hasattr(torch.cuda, 'tunable')):
        return None
    tn = torch.cuda.tunable
    if filename is None:
        os.makedirs(WORK_DIR, exist_ok=True)
        filename = os.path.join(WORK_DIR, f'tunableop_results_dev{device_index}.csv')
        if os.path.exists(SEED) and (tuning is False or not os.path.exists(filename)):
            shutil.copy(SEED, filename)
    tn.enable(True)
    tn.tuning_enable(bool(tuning))
    tn.set_filename(filename)
    if os.path.exists(filename):
        ok = tn.read_file(filename)
        n = len(tn.get_results()) if hasattr(tn, 'get_results') else -1
        if ok is False or n == 0:
            # TunableOp validates the record against the installed hipBLASLt / rocBLAS / ROCm versions and silently ignores it otherwise:
            # the decoder GEMMs then run on the library heuristic (measured 217 instead of 291 frames/s at config 3)
            import warnings
            warnings.warn(f'stereospike_amd.gemm_tuning: the tracked GEMM-algorithm record {SEED} was REJECTED by TunableOp (made with '
                          f'other library versions?): GEMMs fall back to the library defaults; re-tune with tools/tune_gemms.sh '
                          f'(bench.py --gemm-tuning 2)', RuntimeWarning)
    return filename
