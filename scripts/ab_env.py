"""A/B of an environment switch on the cnot3 headline evaluation, both variants in ONE process on the same box, interleaved:
    python3 scripts/ab_env.py QGD_PATHS=inv_panels   (or any VAR=value, e.g. QGD_LIB_PATH=scripts/ubench/bin/libqgd_x.so from scripts/build_variant.sh)
Each variant: its own handle (the switch is read when the library first needs it, so the variants run in child processes),
60 untimed evaluations, then the median of 300 timed ones; the gradient of the two variants is compared bit for bit."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time, hashlib
    import numpy as np, torch
    from __graft_entry__ import import_package
    import cases
    qgd = import_package()
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=550, tf=550.0)
    dp = qgd.DeviceProblem(prob, 8); dp.set_controls(ctrl); dp.set_target(target); dp.set_timing(0)
    for _ in range(60): g, o = dp.discrete_adjoint(pcof)
    ts = []
    for _ in range(300):
        t1 = time.perf_counter(); g, o = dp.discrete_adjoint(pcof); ts.append(time.perf_counter() - t1)
    dp.set_timing(1)
    acc = {}
    for _ in range(5):
        dp.discrete_adjoint(pcof)
        for k, v in dp.timings().items(): acc[k] = acc.get(k, 0) + v / 5 * 1e3
    print(json.dumps({"median_us": float(np.median(ts) * 1e6), "min_us": float(np.min(ts) * 1e6), "sha": hashlib.sha1(g.tobytes()).hexdigest()[:12],
                      "phases_us": {k: round(v, 1) for k, v in acc.items()}}))
    sys.exit(0)
var, _, val = sys.argv[1].partition("=")      # VAR (on = "1") or VAR=value
for rep in range(2):
    for on in (False, True):
        env = dict(os.environ)
        env.pop(var, None)
        if on: env[var] = val or "1"
        out = subprocess.run([sys.executable, __file__, "--child", var], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(var, "on " if on else "off", line[-1] if line else out.stderr[-500:], flush=True)
