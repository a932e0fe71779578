set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/bench_train.py --steps 6 > $O/train_step.json 2> $O/train_step.log; cat $O/train_step.json | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -q -k "training or checkpoint or ddp or weights_reloaded or denoise_step_vs_reference_golden or vae or clip" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
