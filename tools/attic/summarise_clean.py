#!/usr/bin/env python3
"""gpurun_out/r3/clean/run_NN.log (tools/clean_loop.sh) -> profiles/r03_clean_runs.md + the logs themselves under
profiles/r03_clean/ (one per full run of `pytest -m gpu`; each names its tests on stderr as they start)."""
import glob, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'r3', 'clean')
dst = os.path.join(ROOT, 'profiles', 'r03_clean')
os.makedirs(dst, exist_ok=True)
rows, clean = [], 0
for f in sorted(glob.glob(os.path.join(src, 'run_*.log'))):
    txt = open(f, errors='replace').read()
    m = re.findall(r'(\d+) passed[^\n]*', txt)
    tail = re.findall(r'=* ?([^\n=]*(?:passed|failed|error)[^\n=]*) ?=*\n?', txt)
    n_tests = len(re.findall(r'^\[snk-test\] ', txt, flags=re.M))
    fault = re.findall(r'Memory access fault[^\n]*|Aborted[^\n]*|core dumped[^\n]*|HSA_STATUS_ERROR[^\n]*', txt)
    retried = len(re.findall(r'multi-rank child died with a signal', txt))
    ok = bool(m) and 'failed' not in (tail[-1] if tail else 'failed') and not fault and not retried
    clean += ok
    rows.append((os.path.basename(f), tail[-1].strip() if tail else '(no summary line: the process died)', n_tests, ('; '.join(fault[:2]) or '-') + (' (%d child retried)' % retried if retried else ''), 'clean' if ok else 'NOT CLEAN'))
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
head = subprocess.run(['git', 'log', '-1', '--format=%h %s', '--', 'snickery_amd/csrc', 'snickery_amd/engine.py', 'tests'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
lines = ['# Round 3: consecutive full runs of `python -m pytest tests/ -q -m gpu` on MI355X boxes', '',
         'Last commit touching library or tests: `%s` (see the text below for which runs used which).  `tools/clean_loop.sh` runs the whole GPU suite again and again in fresh Python' % head,
         'processes (ten runs per box, a fresh box per ten); every log names each test on stderr before it starts (`[snk-test] <nodeid>`),',
         'so a process abort would read as the last test named + the runtime\'s message.  **%d of %d runs clean.**' % (clean, len(rows)),
         'Run 01 (the first run on its box) failed ONE test: `test_bench_multi_rank_one_gpu[2-0-...]` starts bench.py under',
         '`torch.distributed.run` with two ranks SHARING the box\'s one GPU, and rank 1 of that child process died with SIGABRT; the test',
         'kept only the last 2 000 characters of the child\'s stderr (the launcher\'s summary), so the message is lost.  The pytest process',
         'itself -- the thing that died in round 2\'s driver run -- went on and finished.  240 further runs of those four tests alone',
         '(`tools/dist_loop.sh 60`) and the 30 full runs after it did not reproduce it.  Since run 21 the tests keep what the child said and',
         'run a child that was killed by a signal once more, with the first report on stderr (a run with such a retry is counted NOT clean here).', '',
         '| log (profiles/r03_clean/) | pytest summary | tests started | runtime faults | |', '|---|---|---|---|---|']
lines += ['| %s | %s | %d | %s | %s |' % r for r in rows]
open(os.path.join(ROOT, 'profiles', 'r03_clean_runs.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:6] + lines[-3:]))
