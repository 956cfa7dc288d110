"""gpurun_out/parity_r06.json (written by tests/conftest.py at the end of a `pytest -m gpu` session: what every parity comparison of the
suite MEASURED, next to what it asserted) -> a markdown table, one row per assertion site.   usage: python tools/parity_summary.py [in] > profiles/r06_parity_margins.md"""
import collections
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_r06.json"
d = json.load(open(path))
by = collections.OrderedDict()
for x in d["records"]:
    by.setdefault((x["site"], x["kind"], x["name"]), []).append(x)
print("# Observed parity margins (`pytest -m gpu`, one MI355X session; exit status %d, %d comparisons)\n" % (d["exitstatus"], len(d["records"])))
print("Every `-m gpu` parity comparison records what it measured (tests/gpu_utils.py); this is the summary of one session, the worst value per")
print("assertion site over the parametrised cases that reach it.  Asserted bounds are held to <= 3x these values (SURVEY 8(c)'s own bound where")
print("that is tighter than needed: colour 1e-5 max(1, |x|), gradients rel-L2 1e-4 on the small scenes).\n")
print("| site | what | cases | observed (worst) | asserted |")
print("|---|---|---|---|---|")
fmt = lambda v: "%.2g" % v
for (site, kind, name), xs in by.items():
    test = xs[0]["test"].split("::")[-1].split("[")[0]
    o = lambda k: max(x["observed"][k] for x in xs)
    a = xs[0]["asserted"]
    if kind == "img_close":
        obs = "max %s, scaled max %s, beyond tol: %s of entries" % (fmt(o("max_err")), fmt(o("max_scaled_err")), fmt(o("frac_over_tol")))
        asr = "%s max(1,abs x), bad <= %s, hard %s" % (fmt(a["tol"]), fmt(a["max_bad_frac"]), a["hard"])
    elif kind == "frac_close":
        obs = "max %s, p99.99 %s, beyond tol: %s, beyond 1e-5 max(1,abs x): %s" % (fmt(o("max_err")), fmt(o("p9999_err")), fmt(o("frac_over_asserted")), fmt(o("frac_over_survey_1e-5")))
        asr = "%s + %s abs x, bad <= %s, hard %s" % (fmt(a["atol"]), fmt(a["rtol"]), fmt(a["max_bad_frac"]), a["hard_atol"])
    elif kind == "grad_close":
        obs = "rel-L2 trimmed %s, all %s" % (fmt(o("rel_l2_trimmed")), fmt(o("rel_l2_all")))
        asr = "trimmed <= %s, all <= %s" % (fmt(a["tol_trim"]), fmt(max(x["asserted"]["tol_all"] for x in xs)))
    else:
        obs = "rel-L2 %s" % fmt(o("rel_l2"))
        asr = "(in the assert at that line)"
    print("| `%s` %s | %s %s | %d | %s | %s |" % (site, test, kind, name, len(xs), obs, asr))
