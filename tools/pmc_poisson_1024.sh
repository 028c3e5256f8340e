cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=gpurun_out/pmc1024_$i; rm -rf $d
  timeout -s KILL 170 rocprofv3 --pmc $set -d $d -o r -- python3 tools/poisson_only.py 1024 1 > gpurun_out/pmc1024_$i.log 2>&1 || exit 1
  python3 tools/rocprof_summary.py counter $(find $d -name '*.db' | head -1) gpurun_out/pmc1024_$i.csv > /dev/null
  rm -rf $d; echo "pass $i done"
done
grep -h "k_fft_z_conv\|k_fft_y_fwd\|k_fft_y_inv" gpurun_out/pmc1024_*.csv
