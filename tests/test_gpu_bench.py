"""bench.py end to end on the GPU box, small: what the driver does at N = 1 -- run it, take the LAST stdout line, parse it.
(r05's record was 22.9 KB and came back `parsed: null`; the N > 1 launches are in test_gpu_dist.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_last_stdout_line_of_bench_is_the_compact_record(tmp_path):
    detail = tmp_path / 'detail.json'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--units', '131072', '--frames', '200',
           '--utts', '8', '--candidates', '50', '--cpu-sample-frames', '40', '--no-cpu-all-cores', '--no-variants', '--no-shapes', '--no-greedy',
           '--detail-out', str(detail)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip().splitlines()[-1]
    assert len(last) < 6144
    js = json.loads(last)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline', 'summary'):
        assert key in js, key
    assert js['n_gpus'] == 1 and js['steps'] == 3 and js['warmup'] == 1 and js['value'] > 0 and js['vs_baseline'] is None
    assert 'workload' in js['config'] and 'host -> HBM' in js['config']['inputs'] and js['config']['steps_in_flight'] in (2, 3)
    ro = js['roofline']
    assert ro['kernel'] == 'join_lb2_kernel' and ro['bound'] == 'hbm' and 0.0 < ro['frac'] <= 1.0 and ro['avg_launch_ms'] > 0
    assert js['roofline_check'] == 'every frac within [0, 1]'
    cb = js['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and cb['gpu_matches_cpu_path'] is True and cb['gpu_matches_cpu_candidates'] is True
    assert js['summary']['tripwires']['join_bound_violations'] == 0 and js['summary']['tripwires']['gpu_matches_f32_prefilter'] is True
    full = json.loads(detail.read_text())
    assert full['value'] == js['value'] and 'stages_ms_per_step' in full and 'other_rooflines' in full
