"""profiles/traffic.json from the PMC summaries of scripts/gpu_profile_r04.sh (gpurun_out/r04/prof/pmc_<size>_<counter>.txt).
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE counts half of a wide coalesced read on gfx950
(MI355X_MICROARCH.md, HBM; calibrated on k_native_hash / k_verlet in profiles/r01).  Records the csrc digest the numbers
were measured on; bench.py flags the traffic figure as stale when the kernels have changed since."""
import hashlib, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04", "prof")
SCOPE = {"k_collide_direct": "native/collide+verlet", "k_os_pass<false": "sort/onesweep", "k_native_hash": "native/hash"}
SIZES = {"1M": 1_000_000, "100M": 100_000_000}

def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.split(" dispatches")[0].strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+per-dispatch\s+([0-9.]+)", line)
            if m:
                out[cur][m.group(1)] = float(m.group(2))
    return out

def digest():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpu-physics-engine_amd", "csrc")
    for f in ("gpe_internal.h", "k_native.hip", "k_onesweep.hip"):     # as bench.py kernel_source_digest
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

native, sq = {}, {}
for tag, n in SIZES.items():
    f, w = parse(os.path.join(src, "pmc_%s_FETCH_SIZE.txt" % tag)), parse(os.path.join(src, "pmc_%s_WRITE_SIZE.txt" % tag))
    i, wt = parse(os.path.join(src, "pmc_%s_SQ_INSTS_VALU.txt" % tag)), parse(os.path.join(src, "pmc_%s_SQ_WAIT_ANY.txt" % tag))
    for kern, scope in SCOPE.items():
        kf = [k for k in f if kern in k]
        if not kf:
            continue
        k = kf[0]
        native.setdefault(scope, {})[str(n)] = int((2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024)
        c = dict(i.get(k, {})); c.update(wt.get(k, {}))
        sq.setdefault(scope, {})[str(n)] = {kk: int(v) for kk, v in c.items()}
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
out = {
    "_about": "HBM bytes per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes, profiles/r04/final_pmc_*): "
              "(2 x FETCH_SIZE + WRITE_SIZE) x 1024.  The factor 2 on FETCH_SIZE is MI355X_MICROARCH.md's gfx950 correction, "
              "calibrated in round 1 on kernels of known traffic (k_verlet reads 20 B/particle and reports FETCH_SIZE = 10.0, "
              "k_native_hash reads 8 and reports 4.0; WRITE_SIZE matches their stores exactly).  `sq`: SQ counters per launch "
              "from the same profile set (quad-cycle units for SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_*; GRBM_GUI_ACTIVE is "
              "summed over the 8 XCDs).  Written by scripts/update_traffic_json.py.",
    "_measured_on": {"commit": commit, "csrc_sha16": digest(), "kernel": "k_collide_direct<32, 928, false>"},
    "native": native, "sq": sq, "compat": old.get("compat", {}),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["native"], indent=1)); print(out["_measured_on"])
