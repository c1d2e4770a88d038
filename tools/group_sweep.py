"""Headline (C2) by pipeline arrangement: python tools/group_sweep.py  -> one line per (group, lanes, scl_streams)."""
import json, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = [(0, 7, 2), (16, 4, 2), (24, 4, 2), (12, 4, 2), (16, 3, 3), (32, 4, 2), (16, 5, 2)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[1].split(",")]
for g, lanes, ns in cfgs:
    r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--legs", "none", "--no-cpu-baseline", "--group", str(g), "--lanes", str(lanes), "--front-lanes", str(lanes),
                        "--scl-streams", str(ns), "--steps", os.environ.get("STEPS", "240")], capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"group={g:3d} lanes={lanes} scl_streams={ns}: {j['value'] / 1e6:.3f} M frames/s  ms/step {j['ms_per_step']:.4f}  lone batch {j['config']['single_batch_latency_ms']:.2f} ms  ok={j['config']['sync_offsets_ok']} listed={j['config']['frames_through_list_decoder']}", flush=True)
    except Exception as e:
        print("failed", g, lanes, ns, r.stderr[-800:], flush=True)
