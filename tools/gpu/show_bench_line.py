import sys, json
d = json.loads(sys.stdin.read())
print("steps", d["steps"], "value %.3e" % d["value"], "us/step %.2f" % (1e3 * d["ms_per_step"]), "deg %.2f / region %.2f" % (d["config"]["mean_degree"], d["config"]["mean_degree_over_timed_region"]), {k: "%.3e" % v["value"] for k, v in d["paths"].items()}, "median_of", d.get("value_median_of"))
