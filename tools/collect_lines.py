import json, os, sys
src = "gpurun_out/final"
names = {"c3": "c3", "c2": "c2", "c4": "c4", "c5": "c5", "c5b_1m": "c5b_1m", "c5b_8m": "c5b_8m", "q1": "q1", "q1h": "q1h", "p2": "p2", "p2s": "p2s", "p2t": "p2t",
         "c5_bdf2_vmax0.5": "c5_bdf2_vmax0.5", "c5b_1m_remove_p_mean1": "c5b_1m_remove_p_mean1"}
for n in ("n1_c3", "n2_c3", "n4_c3", "n1_c4", "n2_c4", "n4_c4", "n1_c5b", "n2_c5b", "n4_c5b"):
    names[n] = n + "_shared_gpu"
for k, v in names.items():
    f = os.path.join(src, k + ".json")
    if not os.path.exists(f):
        continue
    txt = open(f).read().strip().splitlines()
    if not txt:
        print("empty", f); continue
    try:
        l = json.loads(txt[-1])
    except Exception as e:
        print("bad", f, e); continue
    json.dump(l, open("profiles/r04_bench_line_%s.json" % v, "w"))
    print(v, l["value"], l.get("krylov_its_per_step"))
