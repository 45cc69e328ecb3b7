export TMPDIR=/tmp
for task in moment_retrieval moment_segmentation; do
  python tools/joint_profile.py $task
  ( cd /tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o x -- python $GRAFT_REPO_ROOT/tools/joint_profile.py $task > /dev/null 2>&1 )
  python tools/busy_span.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 0.5
done
