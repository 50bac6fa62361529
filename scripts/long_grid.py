"""The reference's longest grid: cnot3 at Hermite order 2 with dt = 1e-4 over tf = 550 -- 5.5 million time steps
(/root/reference/examples/cnot3_optimize_gate.sb:28) -- on ONE MI355X through the window machinery of
qgd_set_memory_budget: one gradient evaluation, its time and window count, the gradient against a
centred directional difference of the objective (two forward evaluations).

    python3 scripts/long_grid.py [nsteps] [order] [budget_GiB]

What bounds this run is printed before anything is allocated: the control basis G (2 (m+1) N_coeff doubles per time point
and control, whole grid, on host and device), the window buffers (the library picks the window: at most 65 000 steps, at
most 70 % of the free device memory), and the host memory it takes to build G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def host_mem_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 2**20
        lim = None
        for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(p):
                v = open(p).read().strip()
                if v.isdigit():
                    lim = int(v) / 2**30
        return avail, lim
    except Exception:
        return None, None


def main():
    nsteps = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_500_000
    order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    m = order // 2
    n_ctrl, n_coef = 3, 60
    g_bytes = 2.0 * (nsteps + 1) * (m + 1) * n_coef * 8 * n_ctrl
    avail, lim = host_mem_gib()
    print(f"grid: {nsteps} steps, order {order}; control basis G = {g_bytes / 2**30:.1f} GiB (host and device); "
          f"host memory available {avail and round(avail)} GiB, cgroup limit {lim and round(lim)} GiB", flush=True)
    cap = min(x for x in (avail, lim) if x) if (avail or lim) else None
    if cap is not None and 2.6 * g_bytes / 2**30 > cap:
        sys.exit(f"not attempted: building G needs ~{2.6 * g_bytes / 2**30:.0f} GiB of host memory (the basis and one copy of one control's "
                 f"tables in flight), {cap:.0f} GiB are available")
    import torch
    from __graft_entry__ import import_package
    import cases
    qgd = import_package()
    t0 = time.perf_counter()
    prob, ctrl, pcof, target = cases.cnot3_case(qgd, nsteps=nsteps, tf=550.0)
    dp = qgd.DeviceProblem(prob, order)
    if budget:
        dp.set_memory_budget(int(budget * 2**30))
    plan = dp.memory_plan()
    print(f"memory plan: {plan['windows']} windows of <= {plan['steps_per_window']} steps, {plan['window_bytes'] / 2**30:.1f} GiB of window buffers "
          f"(free device memory: {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB)", flush=True)
    t1 = time.perf_counter()
    dp.set_controls(ctrl)
    t2 = time.perf_counter()
    print(f"control basis built and uploaded in {t2 - t1:.1f} s (problem + handle {t1 - t0:.1f} s); "
          f"free device memory now {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
    dp.set_target(target)
    g, o = dp.discrete_adjoint(pcof)                       # (first call: one-time launch costs)
    print(f"first evaluation done: infidelity {1 - (o[0]**2 + o[1]**2) / prob.N_ess_levels**2:.6f}, guard {o[2]:.3e}", flush=True)
    ts = []
    for _ in range(2):
        t3 = time.perf_counter(); g, o = dp.discrete_adjoint(pcof); ts.append(time.perf_counter() - t3)
    sec = min(ts)
    d = np.random.default_rng(1).standard_normal(len(pcof)); d /= np.linalg.norm(d)
    # a coefficient acts over the whole gate time (tf = 550): eps * tf is the expansion parameter of the difference quotient,
    # eps = 1e-4 gave 1e-5 relative (truncation); eps = 2e-5 and 1e-5 and their Richardson combination are used
    eps = 2e-5

    def obj(p):
        a, b, gd = dp.eval_forward(p)
        return 1 - (a * a + b * b) / prob.N_ess_levels ** 2 + gd

    t4 = time.perf_counter()
    fd = (obj(pcof + eps * d) - obj(pcof - eps * d)) / (2 * eps)
    t5 = time.perf_counter()
    # the carriers make the objective oscillate on the scale of the coefficients: the truncation error of the centred
    # difference (~eps^2 f''' / 6) is visible at eps = 1e-4.  Second difference at eps / 2 and Richardson's combination
    # (error ~eps^4) separate it from an error of the gradient.
    fd2 = (obj(pcof + 0.5 * eps * d) - obj(pcof - 0.5 * eps * d)) / eps
    fdr = (4.0 * fd2 - fd) / 3.0
    rel, rel2, relr = (abs(x - g @ d) / abs(g @ d) for x in (fd, fd2, fdr))
    print(f"RESULT nsteps {nsteps} order {order}: {sec:.3f} s per gradient evaluation = {nsteps / sec / 1e6:.2f} M timesteps/s in {plan['windows']} windows "
          f"({plan['window_bytes'] / 2**30:.1f} GiB each); forward only {(t5 - t4) / 2:.3f} s; |grad| {np.linalg.norm(g):.6g}; "
          f"directional derivative: adjoint {g @ d:.12g}; centred differences eps={eps:g}: {fd:.12g} (rel. diff. {rel:.2e}), eps={eps / 2:g}: {fd2:.12g} ({rel2:.2e}), "
          f"Richardson: {fdr:.12g} ({relr:.2e})", flush=True)
    dp.close()


if __name__ == "__main__":
    main()
