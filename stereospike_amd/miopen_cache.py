"""MIOpen start-up cost control.  The PyTorch-ROCm wheel ships no pre-compiled MIOpen kernels for gfx950, so the first
call of every distinct conv configuration JIT-compiles its kernels (measured: 5 - 140 s each, ~10 min for the whole
network, profiles/README.md).  MIOpen keeps them in a user cache; pointing that cache at an in-tree, git-ignored
directory lets it travel with the repo snapshot exactly like the built libss_neuron.so, so later processes (and later
GPU boxes) start in seconds.  Must be called before the first conv (ideally before `import torch`)."""
import os

CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'miopen_cache')
# Tracked seed: MIOpen's TEXT find-db / perf-db for this network's conv configurations on gfx950 (which solver won the
# find search, measured on an MI355X; profiles/README.md).  Copied into the cache directory when that has none, so a
# fresh checkout skips the ~2 min search; compiled kernels (the binary .ukdb) are not tracked and are rebuilt on demand.
SEED_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')


def enable_per_rank(local_rank: int, skip_naive_solvers: bool = False):
    """One cache directory per process of a multi-GPU launch (seeded from the shared in-tree one): N ranks running MIOpen's
    find search at the same time would otherwise contend for the same sqlite / text db files."""
    import shutil
    path = f'{CACHE_DIR}_rank{local_rank}'
    if not os.path.isdir(path):
        if os.path.isdir(CACHE_DIR):
            shutil.copytree(CACHE_DIR, path, dirs_exist_ok=True)
        else:
            os.makedirs(path, exist_ok=True)
    return enable(path, skip_naive_solvers)


def enable(path: str = CACHE_DIR, skip_naive_solvers: bool = False):
    """skip_naive_solvers: keep MIOpen's find mode from timing its `naive_conv_*` reference solvers (fp64-accumulating,
    seconds per call at config-3 sizes: they were ~70 of the ~75 s of a find-mode start-up, profiles/r01/).  Only for
    find-mode runs of the NHWC network, where tuned implicit-GEMM / CK solvers always exist."""
    os.makedirs(path, exist_ok=True)
    if os.path.isdir(SEED_DIR):
        import shutil
        for f in os.listdir(SEED_DIR):
            if f.endswith('.txt') and not os.path.exists(os.path.join(path, f)):
                shutil.copy(os.path.join(SEED_DIR, f), os.path.join(path, f))
    os.environ.setdefault('MIOPEN_USER_DB_PATH', path)
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', path)
    if skip_naive_solvers:
        for d in ('FWD', 'BWD', 'WRW'):
            os.environ.setdefault(f'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{d}', '0')
    return path


def enable_hermetic(kernel_cache: str = CACHE_DIR):
    """For the test suite: MIOpen's solver choice must not depend on untracked state.  The user find-db / perf-db directory is a FRESH
    temporary directory seeded only from the tracked `miopen_db/` text records (never the git-ignored `lib/miopen_cache/` that bench.py
    find-mode runs keep appending to); only the compiled-kernel cache (binaries: they change start-up time, not which solver runs) is
    shared with the in-tree directory.  Must be called before the first convolution."""
    import shutil
    import tempfile
    db = tempfile.mkdtemp(prefix='ss_miopen_userdb_')
    if os.path.isdir(SEED_DIR):
        for f in os.listdir(SEED_DIR):
            if f.endswith('.txt'):
                shutil.copy(os.path.join(SEED_DIR, f), os.path.join(db, f))
    os.makedirs(kernel_cache, exist_ok=True)
    os.environ['MIOPEN_USER_DB_PATH'] = db
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', kernel_cache)
    return db
