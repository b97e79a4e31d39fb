#!/bin/bash
# r06al: scale requests in front of the limb requests / with the nt policy (headline shape); the request order of r06ah-aj on configs[2] (asym g32)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06al; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extra --no-parity --no-cpu-baseline --prefill-seqs 0"
run() { name=$1; shift; env "$@" timeout 200 python bench.py $Q > $O/q_$name.json 2> $O/q_$name.err; echo "$name rc=$?"; }
c2() { name=$1; shift; env "$@" timeout 200 python tools/longctx_bench.py 160 160 fp16 0 32 1 11008 32 > $O/c2_$name.json 2> $O/c2_$name.err; echo "c2 $name rc=$? $(cut -c1-230 $O/c2_$name.json)"; }
for rep in 1 2 3; do
  run def_$rep X=1
  run scf_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_scf.so
  run scnt_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_scnt.so
  run scfnt_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_scfnt.so
  c2 def_$rep X=1
  c2 old_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_old.so
  c2 ppa_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_ppa.so
  c2 p2_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_p2.so
  c2 scnt_$rep WOQ_HIP_LIB=$PWD/tools/lib_xq_scnt.so
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06al/q_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        bp = d["roofline"]["by_projection"]
        print("%-12s tok/s %7.1f 128: %7.1f  frac %.4f  qkv %.2f o %.2f gate_up %.2f down %.2f" % (f.split("/")[-1], d["value"], d.get("value_128_steps", 0), d["roofline"]["frac"], bp["qkv"]["us"], bp["o"]["us"], bp["gate_up"]["us"], bp["down"]["us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
