import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, j, g = d["roofline"], d.get("roofline_jacobian_kernel") or {}, d.get("roofline_gram") or {}
def ms(x):
    return (x.get("avg_launch_ms") or 0) * (x.get("launches_per_step") or 1)
print(d["config"]["workload"][:34], "| value %.3e ms/step %.4f | main[%s] %.4f ms frac %.3f | two-kernel %s ms/step: jac %.4f ms frac %s gram %.4f ms frac %s"
      % (d["value"], d["ms_per_step"], r["bound"], ms(r), r["frac"] or 0, d.get("two_kernel_ms_per_step"),
         ms(j), j.get("frac"), ms(g), g.get("frac")))
