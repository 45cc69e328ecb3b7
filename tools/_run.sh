timeout 900 python -m pytest tests/test_gpu_joint.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
bash tools/train_round.sh 2>&1 | grep -E "T= 300|colsum|# step|gemm_f32_kernel"
