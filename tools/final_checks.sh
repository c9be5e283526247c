set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_mix_tab; mkdir -p $OUT
export QUAD_SWEEP_COUNTS=196608 QUAD_SWEEP_KERNELS=quad
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/quad_p1 -o p -- python3 tools/quad_sweep.py k1024 > $OUT/q.csv 2> $OUT/q.err
find $OUT -name "*kernel_trace.csv" -size +2M -delete
for f in $OUT/*_p*/p_counter_collection.csv; do [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_pairing_quad<|k_pairing_quad_wtab<|k_coop_invert<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }; done
unset QUAD_SWEEP_COUNTS QUAD_SWEEP_KERNELS
python3 tools/rates_2048.py > gpurun_out/rates_2048.csv 2> gpurun_out/rates_2048.err
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; tail -3 gpurun_out/gpu_suite.log
SOAK_KEYS=k256,k512,k1024,k1024b,k2048 python3 tools/soak.py 300 123 > gpurun_out/soak6.txt 2> gpurun_out/soak6.err; grep "calls compared" gpurun_out/soak6.txt
