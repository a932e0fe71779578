set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "denoise_step or trajectory or training or graph_replay or view_shard or config3 or unet_vs or sample_then_decode" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
timeout 3000 bash tools/round_artifacts.sh r04 > $O/artifacts.log 2>&1
tail -5 $O/artifacts.log
