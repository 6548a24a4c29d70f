export CILRS_LIB=$PWD/tools/bin/libcilrs_hip_exp.so
for cfg in "0 0" "64 0" "128 0" "192 0" "64 1" "128 1" "192 1"; do set -- $cfg; 
  echo "### SIDE_CUS=$1 mode=$2"
  CILRS_SIDE_CUS=$1 CILRS_SIDE_CU_MODE=$2 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --profile-steps 0 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j=json.loads(l); print('fp32 step', j['ms_per_step'])"
  CILRS_SIDE_CUS=$1 CILRS_SIDE_CU_MODE=$2 timeout -k 10 120 python tools/bf16_train_probe.py resnet50 64 2>/dev/null | grep "bf16:" 
done
