set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -k "training or checkpoint or ddp or weights_reloaded or denoise_step_vs_reference_golden or vae or clip or sample or graph_replay" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
timeout 900 python bench.py --train-step > $O/train_step.json 2> $O/train_step.log; cut -c1-300 $O/train_step.json
