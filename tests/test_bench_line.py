"""The record bench.py prints (no GPU needed): the LAST stdout line is a compact JSON object the driver can parse
(the 22.9 KB line of round 5 came back `parsed: null`), no roofline fraction above its peak is ever
printed as a number, and the multi-rank launch order of main() is what torch.distributed.run needs."""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _detail_record():
    """A full record of an earlier run (profiles/r05_j_bench.json: every leg, every note -- 23 KB)."""
    with open(os.path.join(ROOT, 'profiles', 'r05_j_bench.json')) as f:
        return json.load(f)


def test_the_compact_line_is_small_and_round_trips():
    out = _detail_record()
    assert len(json.dumps(out)) > 20000                      # (the record that did not parse)
    out['roofline_check'] = 'every frac within [0, 1]' if not bench.check_rooflines(out) else 'invalid'
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 6144 and '\n' not in text
    back = json.loads(text)
    assert back == json.loads(json.dumps(line))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'summary'):
        assert key in back, key
    assert 'workload' in back['config'] and 'model' not in back['config']
    for key in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms'):
        assert key in back['roofline'], key
    assert back['roofline']['bound'] in ('hbm', 'mfma') and 0.0 <= back['roofline']['frac'] <= 1.0
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in back['cpu_baseline'], key
    assert back['cpu_baseline']['kind'] in ('port', 'reference')
    # nothing nested deeper than summary -> leg -> field, no free-text notes
    assert 'note' not in json.dumps(back['roofline'])


def test_a_fraction_above_the_peak_is_withheld_not_printed():
    out = {'roofline': {'bound': 'hbm', 'achieved': 11064.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 1.383,
                        'alone': {'achieved': 5800.0, 'peak': 8000.0, 'frac': 0.725}},
           'shapes': [{'roofline': {'bound': 'mfma', 'achieved': -1.0, 'peak': 2516.8, 'frac': -0.1}}]}
    bad = bench.check_rooflines(out)
    assert bad == ['/roofline', '/shapes/0/roofline']
    assert out['roofline']['frac'] is None and out['roofline']['achieved'] is None and 'withheld' in out['roofline']['invalid']
    assert out['roofline']['alone']['frac'] == 0.725


def _calls_in_order(tree, fname):
    """(line, dotted name) of every call inside function `fname`, in source order."""
    fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == fname][0]
    out = []
    for n in ast.walk(fn):
        if isinstance(n, ast.Call):
            f, parts = n.func, []
            while isinstance(f, ast.Attribute):
                parts.append(f.attr)
                f = f.value
            if isinstance(f, ast.Name):
                parts.append(f.id)
            out.append((n.lineno, '.'.join(reversed(parts))))
    return sorted(out)


def test_no_gpu_work_before_the_process_group_exists():
    """Under torch.distributed.run every rank must pick ITS device and open the process group before anything creates a
    context (an engine, an upload, a synchronize): a context on device 0 from every rank is how 8-GPU launches fall over.
    The CPU baseline forks workers, so it must run before the first engine too.  Checked on the source of main()."""
    with open(os.path.join(ROOT, 'bench.py')) as f:
        calls = _calls_in_order(ast.parse(f.read()), 'main')
    first = {}
    for line, name in calls:
        first.setdefault(name, line)
    gpu_first = min(first[n] for n in first if n.startswith('eng.') or n in ('snickery_amd.HipSearchEngine', 'torch.cuda.synchronize'))
    assert first['torch.cuda.set_device'] < first['dist.init_process_group'] < gpu_first
    assert first['cpu_baseline'] < first['snickery_amd.HipSearchEngine']
    # the only torch.cuda call before the group exists is the device selection
    early = [n for line, n in calls if line < first['dist.init_process_group'] and n.startswith('torch.cuda')]
    assert early == ['torch.cuda.set_device']


def test_the_profile_summariser_refuses_a_tree_that_has_moved_on(tmp_path):
    """tools/summarise_round6.py compares the sha256 of every kernel source (and of bench.py), taken by tools/prof_round6.sh before
    its first command, with the tree: counters of another build must not become this build's evidence."""
    import hashlib
    import subprocess
    src = tmp_path / 'r06x'
    src.mkdir()
    good = hashlib.sha256(open(os.path.join(ROOT, 'bench.py'), 'rb').read()).hexdigest()
    (src / 'csrc.sha256').write_text('%s  bench.py\n%s  snickery_amd/csrc/joinlb2_kernels.hip\n' % (good, '0' * 64))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'summarise_round6.py'), str(src), 'r06_x', '--out', str(tmp_path)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'refused' in (r.stderr + r.stdout) and 'joinlb2_kernels.hip' in (r.stderr + r.stdout)
    assert not any(p.name.startswith('r06') and p.suffix == '.json' for p in tmp_path.iterdir())
