"""Summarise the four rocprofv3 --pmc passes of tools/pmc.sh per kernel: launches, mean duration, HBM traffic per launch.
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md §HBM) and both counters are in KiB (rocprofv3 unit)."""
import re
import collections, csv, glob, json, sys

def load(d):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in cnt.items()}

def shorten(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in k:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    name = "".join(out).strip()
    if name.startswith("_Z"):  # a mangled name rocprofv3 left alone: keep it from the kernel's own name on, not its last characters
        m = re.search(r"\d+((?:ln_mlp|hiera_\w+?|im2col_u8|rope|stem_conv|cast|upsample2|hyper_mask)_?kernel\w*)", name)
        if m:
            return m.group(1)[:72]
    return name[-72:]


def main(root, out=None):
    rows = {}
    for p in ("p1", "p2", "p3", "p4"):
        acc, cnt = load(f"{root}/{p}")
        for k, c in acc.items():
            r = rows.setdefault(k, {"launches": cnt[k]})
            for n, v in c.items():
                r[n] = v / cnt[k]
    res = {}
    for k, r in rows.items():
        short = shorten(k)
        rd = 2 * r.get("FETCH_SIZE", 0) * 1024
        wr = r.get("WRITE_SIZE", 0) * 1024
        wc = r.get("SQ_WAVE_CYCLES", 0) or 1
        res[short if short not in res else short + "#" + str(len(res))] = dict(launches=r["launches"], read_MB=rd / 1e6, write_MB=wr / 1e6,
            wait_any=r.get("SQ_WAIT_ANY", 0) / wc, wait_inst=r.get("SQ_WAIT_INST_ANY", 0) / wc, active=r.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            mfma_busy=r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(r.get("SQ_BUSY_CYCLES", 1), 1),
            lds_conflict=r.get("SQ_LDS_BANK_CONFLICT", 0) / max(r.get("SQ_LDS_IDX_ACTIVE", 1), 1), grbm=r.get("GRBM_GUI_ACTIVE", 0))
    tot = sorted(res.items(), key=lambda kv: -(kv[1]["read_MB"] + kv[1]["write_MB"]) * kv[1]["launches"])
    print(f"{'kernel':72s} {'n':>6s} {'rd MB':>9s} {'wr MB':>9s} {'waitany':>7s} {'waitinst':>8s} {'active':>6s} {'ldsconf':>7s}")
    for k, v in tot[:40]:
        print(f"{k:72s} {v['launches']:6d} {v['read_MB']:9.2f} {v['write_MB']:9.2f} {v['wait_any']:7.2f} {v['wait_inst']:8.2f} {v['active']:6.2f} {v['lds_conflict']:7.3f}")
    if out:
        json.dump(res, open(out, "w"), indent=1)

if __name__ == "__main__":
    main(*sys.argv[1:])
