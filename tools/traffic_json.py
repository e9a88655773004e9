#!/usr/bin/env python3
"""HBM traffic per launch and kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 128-byte requests at 64 B, so it is doubled; WRITE_SIZE
is used as reported.  Both factors are calibrated on softmax_rows, whose traffic is known exactly (reads rows*cols*4
bytes once).  usage: traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> [key=value ...] > traffic.json
key=value pairs (e.g. _workload=sintel _clips=8 _precision=f16x2 _corr_dtype=f16) are stored verbatim: bench.py only
attaches these bytes to a run of the same configuration."""
import collections, csv, json, re, sys

FAMILY = [(r"corr_build|split_pack|pack_f16", "corr_build"), (r"corr_lookup", "corr_lookup"),
          (r"flash_project_v", "gma_project_v"), (r"ffn_pair_kernel", "ffn_pair"), (r"sk_tail_kernel", "sk_tail"), (r"gma_pv_kernel", "gma_stored"), (r"temporal_block_kernel", "temporal_block"), (r"mask_upsample_kernel", "mask_upsample"),
          (r"gma_flash|flash_pack_v", "gma_flash"), (r"flash_pack_qk", "flash_pack_qk"),
          (r"gemm_f16x3_mfma<[^>]*, 1, [13], (true|false)>", "gemm_attn"), (r"gemm_f|gemm_bdirect|gemm_bstat", "gemm"),
          (r"splitk_epilogue", "gemm"), (r"splitk_combine", "splitk_combine"), (r"dwconv_mfma_kernel<7", "dwconv7"), (r"dwconv_mfma", "dwconv15"),
          (r"dwconv_res_gelu_kernel<15>", "dwconv15"), (r"dwconv_res_gelu_kernel<7>", "dwconv7"),
          (r"softmax_rows", "softmax_rows"), (r"layernorm", "layernorm"), (r"temporal_attn", "temporal_attn"),
          (r"upsample", "upsample_flow"), (r"flow_update", "flow_update"), (r"context_split", "context_split"),
          (r"pack_koct", "pack_koct")]

def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam
    return None

PACK_V_SEEN = False      # (round 5: with sf_gma_flash_project_v the aggregate call launches the fused kernel alone)


def collect(path, counter):
    global PACK_V_SEEN
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        if "flash_pack_v" in r["Kernel_Name"]:
            PACK_V_SEEN = True
        f = family(r["Kernel_Name"])
        if f:
            tot[f] += float(r["Counter_Value"])
            cnt[f] += 1
    return tot, cnt

def main():
    ft, fc = collect(sys.argv[1], "FETCH_SIZE")
    wt, wc = collect(sys.argv[2], "WRITE_SIZE")
    # split_pack + corr_build_dma are two launches of one sf_corr_build_pyramid call: count calls, not launches
    out = {"_note": "KiB per launch (mean). fetch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section), "
                    "write = WRITE_SIZE; calibrated on softmax_rows (known bytes)."}
    for f in sorted(set(ft) | set(wt)):
        nf, nw = max(fc[f], 1), max(wc[f], 1)
        if f == "corr_build" or (f == "gma_flash" and PACK_V_SEEN):     # pack + main kernel = two launches of one C-ABI call
            nf, nw = max(nf // 2, 1), max(nw // 2, 1)
        out[f] = {"fetch_kib_per_launch": round(2.0 * ft[f] / nf, 1), "write_kib_per_launch": round(wt[f] / nw, 1),
                  "fetch_size_raw_kib": round(ft[f] / nf, 1), "launches_profiled": nf}
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        out[k] = int(v) if (v.isdigit() and k != "_csrc_sha") else v
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
