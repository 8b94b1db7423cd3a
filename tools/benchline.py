import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["config"]["workload"][:30], "value %.3e ms/step %.4f jac %.4f ms frac %.3f gram %.4f ms frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]*d["roofline"].get("launches_per_step",1), d["roofline"]["frac"], d["roofline_gram"]["avg_launch_ms"]*d["roofline_gram"].get("launches_per_step",1), d["roofline_gram"]["frac"]))
