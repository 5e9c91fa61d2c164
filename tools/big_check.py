import sys, hashlib
sys.path.insert(0, '.')
import lpr_381_group_v22_amd as pkg
eng = pkg.Engine(0)
for (m, n) in [(6144, 12288), (12000, 2000)]:
    a = pkg.Tableau.synthetic(eng, m, n, 3)
    b = pkg.Tableau.synthetic(eng, m, n, 3)
    ra = a.solve(max_pivots=200)
    rb = b.solve(max_pivots=200, block=1)
    ok = (ra.status == rb.status and ra.pivots == rb.pivots and ra.z == rb.z and
          (a.pivot_log() == b.pivot_log()).all() and
          hashlib.sha256(a.read().tobytes()).hexdigest() == hashlib.sha256(b.read().tobytes()).hexdigest())
    import time
    t0 = time.perf_counter(); r2 = a.solve(max_pivots=2048); dt = time.perf_counter() - t0
    print(m, n, 'block', ra.block, 'same as one-pivot path:', ok, 'status', ra.status, 'pivots', ra.pivots, 'rate', round(r2.pivots / dt), 'pivots/s')
    a.destroy(); b.destroy()
eng.close()
