# host tail old vs new (two builds of libbpmi, BPMI_LIB), n = 2^16: synchronous and two in flight, alternating
for i in 1 2 3; do
  for lib in libbpmi_oldtail.so libbpmi.so; do
    echo "== $lib"; BPMI_LIB=$PWD/python-bulletproofs_amd/$lib R5_CONFIGS="default:" R5_ROUNDS=2 python tools/r05_ab_mid.py 65536 2>&1 | grep "##"
  done
done
