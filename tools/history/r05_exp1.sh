set -x
mkdir -p gpurun_out/r05_exp1
python tools/ab_option.py split 0 1 16 17 > gpurun_out/r05_exp1/split.txt 2>&1
python tools/ab_option.py hist_threads 0 256 16 17 > gpurun_out/r05_exp1/hist_threads.txt 2>&1
python tools/ab_option.py reduce_epl 0 8 16 > gpurun_out/r05_exp1/epl.txt 2>&1
python tools/ab_option.py window_bits 0 14 16 17 > gpurun_out/r05_exp1/c14.txt 2>&1
tail -n 8 gpurun_out/r05_exp1/*.txt
nproc; lscpu | head -20
