#!/usr/bin/env python3
"""Turns the rocprofv3 runs of tools/prof_all.sh into the files kept under profiles/:
   python3 tools/make_traffic.py gpurun_out/prof_final <tag>
writes, per profiled workload w, profiles/<tag>_traffic_<w>.json and profiles/<tag>_<w>_kernel_stats.csv.

HBM bytes per launch of a kernel (largest dispatch of that kernel in the run):
  read  = 128 * RDREQ_128B + 64 * (RDREQ - RDREQ_32B - RDREQ_128B) + 32 * RDREQ_32B
          (on gfx950 FETCH_SIZE tallies every request at 64 B and so reads half of a 128-B request;
           the per-size request counters give the corrected figure; FETCH_SIZE is kept raw beside it)
  write = WRITE_SIZE * 1024  (= 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B) here)
"""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # csrc_digest: names the sources the profiled library was built from

src, tag = sys.argv[1], sys.argv[2]


def largest(path):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Dispatch_Id"])
        by[key][r["Counter_Name"]] = float(r["Counter_Value"])
        by[key]["_grid"] = int(r["Grid_Size"])
    best = {}
    for (k, _), v in by.items():
        if k not in best or v["_grid"] > best[k]["_grid"]:
            best[k] = v
    return best


for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    w = os.path.basename(wdir)
    if not os.path.isdir(os.path.join(wdir, "rdreq")):
        continue

    def f(name):
        return largest(glob.glob(f"{wdir}/{name}/**/*counter_collection.csv", recursive=True)[0])

    rd, wr, fe, wq = f("rdreq"), f("write"), f("fetch"), f("wrreq")
    sq = f("sq") if os.path.isdir(os.path.join(wdir, "sq")) else {}
    out = {}
    for k in rd:
        if not k.startswith("sg::k_"):
            continue
        n128, n32, n = rd[k]["TCC_EA0_RDREQ_128B_sum"], rd[k]["TCC_EA0_RDREQ_32B_sum"], rd[k]["TCC_EA0_RDREQ_sum"]
        read = 128 * n128 + 32 * n32 + 64 * (n - n128 - n32)
        write = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        out[k.replace("sg::", "")] = {
            "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes": read + write,
            "FETCH_SIZE_KB_raw": fe.get(k, {}).get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": wr.get(k, {}).get("WRITE_SIZE"),
            "TCC_EA0_RDREQ": n, "TCC_EA0_RDREQ_128B": n128, "TCC_EA0_RDREQ_32B": n32,
            "TCC_EA0_WRREQ": wq.get(k, {}).get("TCC_EA0_WRREQ_sum"), "TCC_EA0_WRREQ_64B": wq.get(k, {}).get("TCC_EA0_WRREQ_64B_sum"),
            "TCC_HIT": wq.get(k, {}).get("TCC_HIT_sum"), "TCC_MISS": wq.get(k, {}).get("TCC_MISS_sum"), "grid": rd[k]["_grid"]}
        if k in sq:  # wave-instructions issued and LDS-array cycles of the same launch
            out[k.replace("sg::", "")]["sq"] = {c: v for c, v in sq[k].items() if c != "_grid"}
    json.dump({"workload": w, "csrc_sha1": bench.csrc_digest(),
               "command": f"rocprofv3 --pmc <one counter group per run> -- python3 bench.py --workload {w} --steps 1 --warmup 0 --no-cpu-baseline --no-also",
               "kernels": out}, open(f"profiles/{tag}_traffic_{w}.json", "w"), indent=1)
    stats = glob.glob(f"{wdir}/trace/**/*kernel_stats.csv", recursive=True)
    if stats:
        shutil.copy(stats[0], f"profiles/{tag}_{w}_kernel_stats.csv")
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:4]
    print(w, json.dumps({k: round(v["hbm_bytes"] / 1e9, 3) for k, v in top}))
