timeout 900 python -m pytest tests/test_gpu_joint.py -m gpu -q -x -k "moment or segmentation or retrieval or cache" 2>&1 | tail -3
for t in moment_retrieval moment_segmentation; do python tools/joint_profile.py $t 2>&1 | tail -1; python tools/joint_profile.py $t 2>&1 | tail -1; done
