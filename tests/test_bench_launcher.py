"""bench.py's multi-GPU entry (BASELINE configs[3]/[4]): `--gpus N` must either run N ranks or fail
loudly - never print a line that says n_gpus: 1 for an N-GPU request.  CPU-only checks of the launcher logic."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR')}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_n_devices_refuses_instead_of_reporting_one_gpu():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(['--gpus', str(max(n, 2))])
    assert r.returncode == 2, (r.stdout, r.stderr)
    assert 'refusing' in r.stderr and '"metric"' not in r.stdout


def test_launcher_world_size_must_match_gpus():
    r = _run(['--gpus', '4'], env=dict(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1'))
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr and '"metric"' not in r.stdout


def test_launcher_parent_never_imports_the_product_or_touches_hip():
    """The parent of a self-launched N-rank run only counts devices: the product (whose library load and first call
    initialise HIP) is imported by load_product(), which main() reaches only in a rank process."""
    src = open(BENCH).read()
    head = src[:src.index('def load_product')]
    assert 'insenticap_model_amd' not in head.replace('from insenticap_model_amd import Captioner as Cap_', '')
    main = src[src.index('def main():'):src.index('def run(args):')]
    assert main.index('launch_ranks(args)') < main.index('load_product()')
    launch = src[src.index('def launch_ranks'):src.index('def main():')]
    assert 'torch.distributed.run' in launch and 'os.exec' not in src and 'is_available()' not in launch
