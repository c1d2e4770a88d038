"""The driver's invocation (--steps 20 --warmup 5): headline by pipeline arrangement.  argv: steps, then configs group:front:scl ..."""
import json, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "20"
cfgs = [c.split(":") for c in sys.argv[2:]] or [["16", "4", "2"], ["10", "4", "2"], ["8", "4", "3"], ["7", "4", "3"], ["6", "4", "3"], ["5", "3", "4"]]
for g, fl, ns in cfgs:
    vals = []
    for rep in range(3):
        r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--legs", "none", "--no-cpu-baseline", "--steps", steps, "--warmup", "5",
                            "--group", g, "--front-lanes", fl, "--scl-streams", ns], capture_output=True, text=True)
        try:
            vals.append(json.loads(r.stdout.strip().splitlines()[-1])["value"] / 1e6)
        except Exception:
            print("failed", g, fl, ns, r.stderr[-300:], flush=True)
    print(f"steps {steps} group {g} front {fl} scl {ns}: " + " ".join(f"{v:.3f}" for v in vals) + " M frames/s", flush=True)
