import csv, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ce = [i for i, e in enumerate(ev) if e[2].startswith("ce_fwd_kernel")]
c = ce[-2]
# first backbone-backward kernel after loss: the big level-0 final BN apply (un_bn_bwd_apply with many threads) -> use first wgrad on queue != main
nxt = [i for i in range(c, len(ev)) if "spconv_wgrad2_kernel" in ev[i][2] and i > c + 60]
# window: from ce to the first fwd2 dgrad kernel of the backbone (first spconv_fwd2 after 25+ kernels)
end = next(i for i in range(c + 1, len(ev)) if "un_bn_bwd_apply" in ev[i][2] and int(rows[i]["Grid_Size_X"]) > 100000)
t0 = ev[c][0]
print("window %.3f ms, %d kernels" % ((ev[end][0] - t0) / 1e6, end - c))
for i in range(c, end + 1):
    s, e, n, q = ev[i]
    print("%8.1f us  +%6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.replace("void ", "")[:90]))
