#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tests/micro/pmc_calib.hip) -> gpurun_out/pmc_calibration.json
export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/pmc_calib; rm -rf $out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/micro/pmc_calib.hip -o /tmp/pmc_calib || exit 1
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d $out/$ctr -o pmc -- /tmp/pmc_calib > $out/$ctr.log 2>&1
  echo "$ctr rc=$?"
done
cd $root
python - <<PY
import glob, json, sqlite3
known = {"k_read": (1 << 30, 0), "k_write": (0, 1 << 30), "k_rows": (4096 * 6144 * 44, 4096 * 6144 * 8)}
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("$out/%s/**/*.db" % ctr, recursive=True):
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        name_col = [c for c in cols if c in ("kernel_name", "name")][0]
        per = {}
        for kn, cn, val, did in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % name_col):
            if cn == ctr:
                per[(kn, did)] = per.get((kn, did), 0.0) + val
        by = {}
        for (kn, did), v in per.items():
            by.setdefault(kn.replace("void ", "").split("(")[0], []).append(v)
        for kn, vs in by.items():
            res.setdefault(kn, {})[ctr] = sum(vs) / len(vs)
cal = {}
for kn, d in sorted(res.items()):
    base = kn.split("<")[0]
    if base not in known:
        continue
    rd, wr = known[base]
    f, w = d.get("FETCH_SIZE", 0.0) * 1024.0, d.get("WRITE_SIZE", 0.0) * 1024.0
    cal[kn] = {"known_read_bytes": rd, "known_written_bytes": wr, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w,
               "read_factor": round(rd / f, 4) if rd and f else None, "write_factor": round(wr / w, 4) if wr and w else None}
    print(kn.ljust(28), cal[kn])
json.dump({"source": "tests/tools/gpu_pmc_calib.sh: tests/micro/pmc_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
           "--kernel-trace only); factor = known bytes / (counter x 1024)", "kernels": cal}, open("$root/gpurun_out/pmc_calibration.json", "w"), indent=1)
PY
rm -rf $out
