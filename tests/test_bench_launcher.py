"""``python bench.py --gpus N`` must launch itself: with no torchrun environment and N > 1 the
parent -- before importing torch or touching a GPU -- starts N fresh rank processes through
``python -m torch.distributed.run`` on 127.0.0.1, relays rank 0's single JSON line and the children's exit code.

No GPU here: ``--stub`` swaps the training step for a stand-in on CPU tensors and the backend for gloo; launcher,
rendezvous, barrier / max-over-ranks timing and the result line are the real code paths.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          env=env, timeout=timeout)


def test_bench_self_launches_two_ranks_on_gloo():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--stub', '--backend', 'gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                    # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['n_ranks_seen'] == 2          # counted by an all-reduce of ones
    assert line['steps'] == 3 and line['warmup'] == 1 and line['scaling'] == 'weak'
    assert line['config']['global_batch'] == 2 * 64 and line['config']['parallelism'] == 'dp2'
    assert line['value'] > 0 and line['ms_per_step'] > 0
    # the stand-in step averaged the ranks' "gradients" (1 and 2) three times on top of one warm-up increment
    assert abs(line['checksum'] - 1000 * (1 + 3 * 1.5)) < 1e-6


def test_bench_single_rank_does_not_launch():
    r = _run(['--gpus', '1', '--steps', '2', '--warmup', '0', '--stub'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line['n_gpus'] == 1 and line['n_ranks_seen'] == 1


def test_bench_launcher_reports_child_failure():
    # an argument the rank processes reject: the launcher must come back non-zero and print no result line
    r = _run(['--gpus', '2', '--stub', '--backend', 'no-such-backend', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_bench_under_torchrun_world_mismatch_is_an_error():
    r = _run(['--gpus', '4', '--stub'], env_extra={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'rank processes' in (r.stderr + r.stdout)
