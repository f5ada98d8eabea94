"""A/B of the batched W^T refresh (autograd_ops.WeightTransposes) on the cfg3 decoder training step: argv[1] = on | off."""
import runpy
import sys
sys.path.insert(0, "/root/repo")
import labelanything_amd.autograd_ops as A
if sys.argv[1] == "off":
    def get(self, w):
        wt = w.new_empty(w.shape[1], w.shape[0])
        A.L.nhwc_to_nchw(w, 1, w.shape[1], w.shape[0], wt)
        return wt
    A.WeightTransposes.get = get
sys.argv = ["bench.py", "--workload", "cfg3_train", "--no-cpu-baseline", "--no-eager-baseline"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
