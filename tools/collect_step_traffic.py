"""Whole-decode-step HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over an EAGER (--no-graph)
Whisper bench run: every dispatch of a decode-step kernel summed, divided by the number of steps (= dispatches of the
step's last kernel).  Units / gfx950 correction as tools/collect_traffic.py (KiB; FETCH_SIZE doubled).

    python tools/collect_step_traffic.py OUT/f OUT/w profiles/r02/whisper_step_traffic.json"""
import csv, glob, json, sys

DEC = ("dec_linear_kernel", "dec_logits_kernel", "dec_attn_fused_kernel", "dec_attn_kernel", "dec_argmax_reduce_kernel", "dec_embed_kernel", "dec_layers_kernel",
       "dec_attn_v2_kernel")
LAST = "dec_argmax_reduce_kernel"


def load(dirname, counter):
    """counter: a name, or "RDREQ" = bytes rebuilt from the raw request counters TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum
    (64-byte requests minus the 32-byte ones): the same quantity the derived FETCH_SIZE reports in KiB - that derived counter
    hangs rocprofv3 over the eager Whisper run (r02: 400 s timeout, sf.log), its raw inputs do not."""
    f = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    per, steps = {}, 0
    for r in csv.DictReader(open(f)):
        if counter == "RDREQ":
            if r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
                scale = 64.0 / 1024
            elif r["Counter_Name"] == "TCC_EA0_RDREQ_32B_sum":
                scale = -32.0 / 1024
            else:
                continue
            r = dict(r, Counter_Value=float(r["Counter_Value"]) * scale)
            steps_count = r["Counter_Name"] == "TCC_EA0_RDREQ_sum"
        elif r["Counter_Name"] != counter:
            continue
        else:
            steps_count = True
        name = r["Kernel_Name"]
        key = next((k for k in DEC if k in name), None)
        if key is None:
            continue
        if key == "dec_attn_fused_kernel":
            key += "<self>" if "<true" in name or "Lb1" in name else "<cross>"
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        steps += (LAST in name) and steps_count
    return per, steps


fdir, wdir, out = sys.argv[1:4]
f, n1 = load(fdir, "RDREQ" if "raw" in fdir else "FETCH_SIZE")
w, n2 = load(wdir, "WRITE_SIZE")
res = {
    "steps_sampled": [n1, n2],
    "fetch_bytes_per_step": 2 * 1024 * sum(f.values()) / n1,
    "write_bytes_per_step": 1024 * sum(w.values()) / n2,
    "per_kernel_fetch_bytes_per_step": {k: 2 * 1024 * v / n1 for k, v in sorted(f.items())},
    "per_kernel_write_bytes_per_step": {k: 1024 * v / n2 for k, v in sorted(w.items())},
    "method": "rocprofv3 --pmc in separate passes over `bench.py --workload whisper --no-graph` (eager decode launches: counter collection "
              "cannot sample graph replays): reads = FETCH_SIZE, or (directory name contains 'raw') TCC_EA0_RDREQ_sum x 64 B - "
              "TCC_EA0_RDREQ_32B_sum x 32 B, doubled (gfx950); writes = WRITE_SIZE (KiB); all decode-step kernels summed over the run / steps",
}
res["traffic_bytes_per_launch"] = res["fetch_bytes_per_step"] + res["write_bytes_per_step"]  # "launch" = one decode step here
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
