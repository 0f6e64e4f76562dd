"""Development tool: same-box A/B timing of differently built / differently configured libraries.

    python tools/ab_variants.py [--steps 100] [--smoke] name[=ENV=VAL,ENV=VAL...][@lib] ...

Each configuration runs `bench.py --no-cpu-baseline` in a fresh process (the library is loaded once per process):
`@lib` selects build_variants/libgsr_<lib>.so through GSR_LIBRARY_PATH (default: the in-tree library), `ENV=VAL` pairs
are added to the environment.  `--smoke` first checks every configuration against the oracle
(`__graft_entry__.smoke()`), so a variant that miscompiles is not timed.  Prints one table; every JSON line goes to
gpurun_out/ab_<name>.json.  Example:

    python tools/ab_variants.py base ilp@ilp_blend fastexp=GSR_FAST_EXP=1 ilp_fast=GSR_FAST_EXP=1@ilp_blend
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    steps, smoke, extra = "100", False, []
    cfgs = []
    i = 0
    while i < len(args):
        a = args[i]
        if a == "--steps":
            steps = args[i + 1]
            i += 2
            continue
        if a == "--scene":  # (a bench.py option with a non-numeric value)
            extra += [a, args[i + 1]]
            i += 2
            continue
        if a == "--smoke":
            smoke = True
        elif a.startswith("--"):
            extra.append(a)
            if i + 1 < len(args) and not args[i + 1].startswith("--") and "=" not in args[i + 1] and "@" not in args[i + 1] \
                    and args[i + 1].replace(".", "").isdigit():
                extra.append(args[i + 1])
                i += 1
        else:
            cfgs.append(a)
        i += 1
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rows = []
    for c in cfgs:
        lib = None
        if "@" in c:
            c, lib = c.split("@", 1)
        name, _, envs = c.partition("=")
        env = dict(os.environ)
        if envs:
            for kv in envs.split(","):
                k, _, v = kv.partition("=")
                env[k] = v
        if lib:
            env["GSR_LIBRARY_PATH"] = os.path.join(ROOT, "build_variants", f"libgsr_{lib}.so")
        if smoke:
            p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                rows.append((name, None, "SMOKE FAILED: " + (p.stderr.strip().splitlines() or ["?"])[-1]))
                continue
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-configs", "--steps", steps, "--warmup", "10"]
                           + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        line = next((ln for ln in reversed(p.stdout.splitlines()) if ln.startswith("{")), None)
        if p.returncode != 0 or line is None:
            rows.append((name, None, "BENCH FAILED: " + (p.stderr.strip().splitlines() or ["?"])[-1]))
            continue
        j = json.loads(line)
        open(os.path.join(ROOT, "gpurun_out", f"ab_{name}.json"), "w").write(line + "\n")
        rows.append((name, j, ""))
    stages = ("preprocess", "bin", "blend_forward", "blend_backward", "preprocess_backward")
    print(f"{'config':<22}{'it/s':>9}{'ms/step':>9}{'med':>8}{'fwd ms':>8}" + "".join(f"{s[:12]:>13}" for s in stages))
    for name, j, err in rows:
        if j is None:
            print(f"{name:<22}{err}")
            continue
        sm = j["stage_ms"]
        print(f"{name:<22}{j['value']:>9.1f}{j['ms_per_step']:>9.4f}{j['step_ms_gpu']['median']:>8.4f}{j['forward_ms']:>8.4f}"
              + "".join(f"{1e3 * sm[s]:>13.1f}" for s in stages))


if __name__ == "__main__":
    main()
