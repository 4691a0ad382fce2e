"""`python bench.py --gpus N` must itself start N ranks (round-1 finding: the flag was parsed and ignored).

CPU: the launcher + rendezvous + max-over-ranks aggregation + single JSON line, with a sleep standing in for the
kernel (`--rehearse-cpu`).  GPU (one card): the real bench with two gloo ranks that share cuda:0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env=None, timeout=600):
  e = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    e.pop(k, None)
  e.update(env or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True,
                        env=e, timeout=timeout, cwd=ROOT)


def _one_json_line(stdout):
  lines = [l for l in stdout.splitlines() if l.strip()]
  assert len(lines) == 1, stdout
  return json.loads(lines[0])


def test_gpus_flag_alone_starts_the_ranks():
  r = _run(['--gpus', '2', '--rehearse-cpu', '--steps', '4', '--envs', '1000'])
  assert r.returncode == 0, r.stderr[-2000:]
  d = _one_json_line(r.stdout)
  assert d['n_gpus'] == 2 and d['config']['global_envs'] == 2000 and d['config']['env_id_base_of_last_rank'] == 1000
  # max over ranks: rank 1 sleeps 2 ms per step
  assert d['ms_per_step'] >= 2.0
  # the line proves who took part: one record per rank, the process group's own size and backend, distinct devices
  assert [r['rank'] for r in d['ranks']] == [0, 1] and d['ranks'][1]['kernel_ms'] >= 2.0 > d['ranks'][0]['kernel_ms'] >= 1.0
  assert d['distributed'] == {'backend': 'gloo', 'world_size': 2, 'distinct_devices': 2}


def test_ranks_that_share_a_device_are_an_error_outside_a_rehearsal():
  sys.path.insert(0, ROOT)
  import bench
  two = [{'rank': 0, 'pci_bus_id': '0000:05:00'}, {'rank': 1, 'pci_bus_id': '0000:05:00'}]
  with pytest.raises(SystemExit) as e:
    bench.check_distinct_devices(two, allow_shared=False)
  assert '0000:05:00' in str(e.value)
  assert bench.check_distinct_devices(two, allow_shared=True) == 1
  assert bench.check_distinct_devices([two[0], {'rank': 1, 'pci_bus_id': '0000:15:00'}], allow_shared=False) == 2


def test_world_size_that_disagrees_with_the_flag_is_an_error():
  r = _run(['--gpus', '3', '--rehearse-cpu'], env={'RANK': '0', 'WORLD_SIZE': '2'})
  assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr


def test_single_rank_needs_no_launcher():
  r = _run(['--rehearse-cpu', '--steps', '2'])
  assert r.returncode == 0, r.stderr[-2000:]
  assert _one_json_line(r.stdout)['n_gpus'] == 1


@pytest.mark.gpu
def test_two_ranks_on_one_card_real_bench():
  r = _run(['--gpus', '2', '--backend', 'gloo', '--all-on-device0', '--envs', '8192', '--also-envs', '0', '--rollout-envs', '4096', '--rollout-k', '8', '--config4-envs', '4096', '--steps', '30',
            '--warmup', '5', '--no-cpu-baseline', '--stagger'])
  assert r.returncode == 0, r.stderr[-3000:]
  d = _one_json_line(r.stdout)
  assert d['n_gpus'] == 2 and d['config']['envs_per_gpu'] == 8192 and d['config']['global_envs'] == 16384
  assert d['device_errors'] == 0 and d['value'] > 0
  assert d['rollout']['device_errors'] == 0 and d['rollout']['env_steps_timed'] == 24 and d['rollout']['value'] > 0
  assert d['config4']['device_errors'] == 0 and d['config4']['value'] > 0 and d['config4']['rollout']['device_errors'] == 0
  assert 0.0 < d['config']['auto_reset']['env_steps_reset_in_timed_region_frac'] < 0.05
  # two records, both naming the one card (a declared rehearsal: the same PCI bus id is allowed), each with its own kernel time
  assert [r['rank'] for r in d['ranks']] == [0, 1] and d['ranks'][0]['pci_bus_id'] == d['ranks'][1]['pci_bus_id'] is not None
  assert all(r['kernel_ms'] > 0 and r['device'] and r['compute_units'] > 0 for r in d['ranks'])
  assert d['distributed'] == {'backend': 'gloo', 'world_size': 2, 'distinct_devices': 1, 'shared_device_rehearsal': True}
