"""Build profiles/<round>_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
runs, CSV output) of `bench.py --steps 3`.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB; FETCH_SIZE
reports half the bytes of a wide coalesced read stream on gfx950 -> read bytes = FETCH_SIZE*1024*2;
WRITE_SIZE is exact for wide stores.
    python tools/pmc_traffic.py gpurun_out/r01c_fetch gpurun_out/r01c_write profiles/r01_traffic.json"""
import csv, glob, json, os, sys, collections

LAYER_OF = [  # kernel-name fragment -> layer key used by bench.py's timing slots (fcn_skip, 2048x1536)
    ("conv12_ws_kernel", "conv2d_1"),
    ("conv_mfma_kernel<8, 2, 5, 1, 3, 0, 33", "conv2d_1"),
    ("conv_mfma_kernel<8, 2, 5, 1, 3, 0, 97", "conv2d_1"),
    ("conv_mfma_kernel<8, 2, 5, 1, 3, 0, 353", "conv2d_1"),
    ("conv_mfma_kernel<8, 2, 5, 1, 3, 0, 289", "conv2d_1"),
    ("tail_composed_kernel", "conv2d_transpose_4"),
    ("tail_fused2_kernel", "conv2d_transpose_4"),
]


def load(d, counter):
    rows = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py --steps 3 "
                    "--no-cpu-baseline`, averaged per launch (tools/pmc_traffic.py). Units KiB. Per MI355X_MICROARCH.md (HBM): on "
                    "gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read stream -> hbm_read_bytes = "
                    "FETCH_SIZE*1024*2; WRITE_SIZE is exact for wide stores. Keys: bench.py timing-slot names where mapped, "
                    "otherwise the kernel name."}
    for k in sorted(set(fetch) | set(write)):
        if "pseg::" not in k:
            continue
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        key = next((ly for frag, ly in LAYER_OF if frag in k), k[:90])
        res[key] = {"kernel": k[:160], "launches_seen": len(fetch.get(k, [])), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                    "hbm_read_bytes": int(f * 1024 * 2), "hbm_write_bytes": int(w * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res) - 1, "kernels")


if __name__ == "__main__":
    main()
