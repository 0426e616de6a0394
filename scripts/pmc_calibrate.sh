#!/bin/bash
# GPU box: builds and runs scripts/pmc_calibrate.hip under two rocprofv3 PMC passes -> gpurun_out/pmc_calibration.json
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gpurun_out/calib/pmc_calibrate scripts/pmc_calibrate.hip || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/calib/$C -- gpurun_out/calib/pmc_calibrate > gpurun_out/calib/$C.log 2>&1 || { tail -5 gpurun_out/calib/$C.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json
bytes_ = {"read8": 2 << 30, "write8": 2 << 30, "read16": 2 << 30, "write16": 2 << 30, "write24s": (2 << 30) // 24 * 24, "gather128": (2 << 30) // 128 // 4 * 128}
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/calib/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k in bytes_ and r["Counter_Name"] == c:
                val[(k, c)] = float(r["Counter_Value"])
out = {"raw_counter_values": {"%s:%s" % k: v for k, v in val.items()}, "known_bytes": bytes_}
out["bytes_per_unit"] = {"FETCH_SIZE:read8": bytes_["read8"] / val[("read8", "FETCH_SIZE")], "FETCH_SIZE:read16": bytes_["read16"] / val[("read16", "FETCH_SIZE")],
                         "FETCH_SIZE:gather128": bytes_["gather128"] / val[("gather128", "FETCH_SIZE")],
                         "WRITE_SIZE:write8": bytes_["write8"] / val[("write8", "WRITE_SIZE")], "WRITE_SIZE:write16": bytes_["write16"] / val[("write16", "WRITE_SIZE")],
                         "WRITE_SIZE:write24s": bytes_["write24s"] / val[("write24s", "WRITE_SIZE")]}
json.dump(out, open("gpurun_out/pmc_calibration_raw.json", "w"), indent=1)
print(json.dumps(out["bytes_per_unit"], indent=1))
PY
