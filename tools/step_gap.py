"""Step-boundary analysis of a rocprofv3 kernel trace of bench.py: for every optimiser step, the idle time of the main
queue between the last backward kernel and the first forward kernel of the next step, and what the side queue ran then.

  rocprofv3 --kernel-trace -d /tmp/tr -o run --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile
  python3 tools/step_gap.py /tmp/tr
"""
import csv
import glob
import sys
from collections import Counter


def main(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    starts = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
    qcount = Counter(r["Queue_Id"] for r in rows)
    main_q = rows[starts[-1]]["Queue_Id"]
    print("queues:", dict(qcount), "main:", main_q)
    for a, b in zip(starts[-4:-1], starts[-3:]):
        step = rows[a:b]
        t0, t1 = rows[a]["s"], rows[b]["s"]
        mainq = [r for r in step if r["Queue_Id"] == main_q]
        side = [r for r in step if r["Queue_Id"] != main_q]
        busy_main = sum(r["e"] - r["s"] for r in mainq)
        last_main_end = max(r["e"] for r in mainq)
        gap = t1 - last_main_end
        in_gap = [r for r in side if r["e"] > last_main_end]
        print(f"step {(t1 - t0) / 1e6:.3f} ms: main-queue kernel time {busy_main / 1e6:.3f} ms, idle {((t1 - t0) - busy_main) / 1e6:.3f} ms; "
              f"end-of-step gap {gap / 1e3:.1f} us with {len(in_gap)} side kernels: "
              + ", ".join(f"{r['Kernel_Name'][:28]}({(r['e'] - max(r['s'], last_main_end)) / 1e3:.0f}us)" for r in in_gap[:8]))
        # largest idle intervals of the main queue inside the step
        gaps = sorted(((mainq[i + 1]["s"] - mainq[i]["e"], mainq[i]["Kernel_Name"][:30], mainq[i + 1]["Kernel_Name"][:30]) for i in range(len(mainq) - 1)),
                      reverse=True)[:6]
        print("   largest main-queue gaps (us):", [(round(g / 1e3, 1), x, y) for g, x, y in gaps])
        # launch-count view of the step: kernels per queue, the idle time behind each kernel name (what the dependent chains of
        # ten-microsecond launches cost), and the at::native launches inside the step (fills / copies that are NOT build-time)
        allg = [(mainq[i + 1]["s"] - mainq[i]["e"], mainq[i]["Kernel_Name"]) for i in range(len(mainq) - 1)]
        idle_by = Counter()
        for gns, nm in allg:
            idle_by[nm.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][-40:]] += max(gns, 0)
        print(f"   kernels: main {len(mainq)}, side {len(side)}; main-queue gaps: {sum(1 for g_, _ in allg if g_ > 2000)} above 2 us, "
              f"sum {sum(max(g_, 0) for g_, _ in allg) / 1e6:.3f} ms")
        print("   idle behind (ms):", [(k, round(v / 1e6, 3)) for k, v in idle_by.most_common(8)])
        def short(nm):
            return nm.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][-44:]
        busy_by = Counter()
        for r in mainq:
            busy_by[short(r["Kernel_Name"])] += r["e"] - r["s"]
        side_by = Counter()
        for r in side:
            side_by[short(r["Kernel_Name"])] += r["e"] - r["s"]
        print("   main-queue kernel time by kernel (ms):", [(k, round(v / 1e6, 2)) for k, v in busy_by.most_common(12)])
        print("   side-queue kernel time by kernel (ms):", [(k, round(v / 1e6, 2)) for k, v in side_by.most_common(6)], "total", round(sum(side_by.values()) / 1e6, 2))
        native = Counter(r["Kernel_Name"].split("<")[0][-48:] for r in step if "at::native" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"])
        print("   at::native / runtime launches inside the step:", dict(native))


if __name__ == "__main__":
    main(sys.argv[1])
