"""How many workgroups of k_front should start from step matrices pre-built by the tables launch?  The cnot3 headline evaluation
under QGD_PATHS=front_pre2=<q2>[,front_pre1=<q1>] (csrc/qgd_front.h: FrontPre), each configuration in its own process
(scripts/ab_env.py --child: 60 untimed + 300 timed evaluations, median), two rounds, interleaved.
   gpurun -- python scripts/front_pre_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = [("no_front", "no_front"), ("none", "front_nopre"), ("39+39 (default)", ""), ("39+128", "front_pre2=128"), ("39+153", "front_pre2=153"),
           ("39+256", "front_pre2=256"), ("39+256+39", "front_pre2=256,front_pre1=39")]
if len(sys.argv) > 1:
    configs = [(a, a) for a in sys.argv[1:]]
for rep in range(2):
    for name, paths in configs:
        env = dict(os.environ, QGD_PATHS=paths)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_env.py"), "--child", "x"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:]); continue
        d = json.loads(line[-1])
        ph = d["phases_us"]
        print(f"{name:18s} median {d['median_us']:7.1f} us  tables {ph.get('tables', 0):5.1f}  front {ph.get('front', ph.get('inverse', 0) + ph.get('build_LR', 0)):6.1f}  sha {d['sha']}", flush=True)
