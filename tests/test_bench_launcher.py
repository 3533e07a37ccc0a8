"""CPU suite: `python bench.py --gpus N` launches its N ranks itself (one process per GPU over RCCL) and refuses to run with a
mismatching WORLD_SIZE (a bare `--gpus 8` used to run ONE rank and report n_gpus 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launcher_command_is_a_torchrun_of_this_script_with_the_same_arguments():
    b = _bench()
    argv = ['--gpus', '8', '--steps', '5', '--warmup', '2']
    cmd = b.launcher_command(8, argv, {})
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == argv
    # an explicit MASTER_PORT is honoured
    assert b.launcher_command(2, argv, {'MASTER_PORT': '29999'})[b.launcher_command(2, argv, {'MASTER_PORT': '29999'}).index('--master-port') + 1] == '29999'


def test_no_spawn_for_one_gpu_or_inside_a_rank():
    b = _bench()
    assert b.launcher_command(1, ['--gpus', '1'], {}) is None
    assert b.launcher_command(8, ['--gpus', '8'], {'WORLD_SIZE': '8', 'RANK': '3'}) is None
    assert b._early_gpus(['--steps', '3']) == 1 and b._early_gpus(['--gpus=4']) == 4 and b._early_gpus(['--gpus', '2']) == 2


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 4 but WORLD_SIZE=2' in (r.stderr + r.stdout)
