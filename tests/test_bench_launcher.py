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
    r = _run(['--gpus', '64'])                                  # (no box has 64)
    assert r.returncode == 2, (r.stdout, r.stderr)
    assert 'refusing' in r.stderr and '"metric"' not in r.stdout


def test_launcher_world_size_must_match_gpus():
    r = _run(['--gpus', '4'], env=dict(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1'))
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr and '"metric"' not in r.stdout


def test_dry_ranks_two_ranks_spawn_relay_and_exit_code():
    """The whole self-launch path, on the CPU: the parent spawns two ranks through torch.distributed.run, they form a
    group (gloo), all-reduce a token and rank 0 prints the one JSON line, which the parent relays on ITS stdout; a rank
    that exits non-zero makes the parent exit non-zero without a result line."""
    import json
    r = _run(['--dry-ranks', '2'])
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] == 3.0 and 'dry-run' in d['metric']
    bad = _run(['--dry-ranks', '2', '--dry-fail-rank', '1'])
    assert bad.returncode != 0 and '"metric"' not in bad.stdout, (bad.returncode, bad.stdout)
    assert 'run failed' in bad.stderr


def test_the_headline_survives_a_side_measurement_that_fails_or_hangs_under_a_group():
    """The N-rank line is the only multi-GPU measurement there is: once the headline has been timed, a side measurement
    that dies on another rank (the launcher then SIGTERMs rank 0, which may sit in a collective - a C-level wait no Python
    signal handler interrupts) or that never returns must not take the line with it.  bench.HeadlineGuard: wake-up pipe +
    watcher thread + deadline; rank 0 writes the line with a note and leaves non-zero, the parent relays line AND code."""
    import json
    late = _run(['--dry-ranks', '2', '--dry-fail-late-rank', '1'])
    assert late.returncode != 0, (late.stdout, late.stderr[-2000:])
    lines = [x for x in late.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, late.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] == 3.0
    assert 'SIGTERM' in d['extra']['side_measurements_aborted'], d
    hang = _run(['--dry-ranks', '2', '--dry-hang', '--extras-deadline', '3'])
    assert hang.returncode != 0, (hang.stdout, hang.stderr[-2000:])
    lines = [x for x in hang.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, hang.stdout
    d = json.loads(lines[0])
    assert d['value'] == 3.0 and 'deadline' in d['extra']['side_measurements_aborted'], d


def test_the_result_line_is_composed_before_the_side_measurements_start():
    """run(): everything the headline's own measurements determine (roofline entries from the kernel timer, value) is
    computed before the first side measurement, so that HeadlineGuard's thread only formats - no device call from it."""
    src = open(BENCH).read()
    run = src[src.index('def run(args):'):src.index("if __name__ == '__main__':")]
    assert run.index('summ = ops.TIMER.summary()') < run.index('def compose(') < run.index('def side_measurements(')
    assert run.index('def side_measurements(') < run.index('HeadlineGuard(compose')
    comp = run[run.index('def compose('):run.index('def side_measurements(')]
    assert 'torch.cuda' not in comp and 'TIMER' not in comp and 'note is None' in comp     # (cpu_baseline: normal path only)


def test_gpu_count_comes_from_sysfs_or_a_child_never_from_torch_in_the_parent(tmp_path, monkeypatch):
    """count_gpus: KFD topology nodes with simd_count > 0, narrowed by the *_VISIBLE_DEVICES lists; this process (the
    would-be parent of the ranks) does not import torch for it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    root = tmp_path / 'nodes'
    for i, simd in enumerate((0, 1024, 1024, 1024)):           # node 0: the CPU
        (root / str(i)).mkdir(parents=True)
        (root / str(i) / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nlocal_mem_size 0\n' % (16 if simd == 0 else 0, simd))
    real_isdir, real_listdir, real_open = os.path.isdir, os.listdir, open
    kfd = '/sys/class/kfd/kfd/topology/nodes'
    monkeypatch.setattr(bench.os.path, 'isdir', lambda p: True if p == kfd else real_isdir(p))
    monkeypatch.setattr(bench.os, 'listdir', lambda p: real_listdir(str(root)) if p == kfd else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, 'open', lambda p, *a, **k: real_open(str(p).replace(kfd, str(root)), *a, **k))
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    assert 'torch' not in bench.__dict__ or bench.torch is None
    assert bench.count_gpus() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2')
    assert bench.count_gpus() == 2
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '1')
    assert bench.count_gpus() == 1
    assert bench.torch is None                                  # still not imported


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
    assert 'device_count()' not in launch                       # (the count comes from count_gpus: sysfs, or a child)


def test_a_rank_fatal_failure_is_not_swallowed_by_the_side_measurement_loop():
    """Round-4 advisor finding: `bench_xe_train` raised an Exception when a graph capture failed under a process group,
    and the jobs loop turned it into extra['xe_train'] = {'error': ...} and moved on to the next job's collectives while
    the other ranks were still inside this job's.  The failure is a BaseException now (bench.FatalUnderGroup): ordinary
    job failures are still reported under their key, this one leaves the loop - and, in a rank process, the interpreter
    with a non-zero code."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert not issubclass(bench.FatalUnderGroup, Exception)
    extra, ran = {}, []

    def ok():
        ran.append('ok')
        return 1

    def soft():
        raise RuntimeError('a side measurement failed')

    def fatal():
        raise bench.FatalUnderGroup('capture failed under the group')

    def never():
        ran.append('never')
    try:
        bench.run_jobs([('a', ok), ('b', soft), ('c', fatal), ('d', never)], extra)
    except bench.FatalUnderGroup:
        pass
    else:
        raise AssertionError('FatalUnderGroup was swallowed')
    assert extra['a'] == 1 and 'error' in extra['b'] and 'c' not in extra and 'd' not in extra and ran == ['ok']
    # ... and as a process: an uncaught BaseException ends the interpreter non-zero
    code = ("import importlib.util as u; s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); "
            "s.loader.exec_module(m); m.run_jobs([('c', lambda: (_ for _ in ()).throw(m.FatalUnderGroup('x')))], {})" % BENCH)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'FatalUnderGroup' in r.stderr
