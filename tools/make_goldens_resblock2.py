"""tests/golden/resblock2_k*.npz: the reference's ResBlock2 (vits/model/modules.py:225-247) run in this container at a
reduced width -- input, seeded parameters, output and the input / parameter gradients of the probe loss sum(y * r) --
with the oracle's restatement (oracle/vits_oracle.py: resblock2_forward) checked against it.

Run:  python tools/make_goldens_resblock2.py      (only here; /root/reference does not exist on the GPU box)"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

# (importing make_goldens installs the import stubs and puts /root/reference on the path)
from make_goldens import O, close, grads_of, load_seeded, rmodules, rng_tensor, save  # noqa: E402

CASES = [(3, (1, 3), 208), (7, (1, 3), 209), (5, (2, 6), 210)]  # (kernel size, dilations, parameter seed)


def main():
    rng = np.random.default_rng(20241006)
    for k, dil, seed in CASES:
        rb = rmodules.ResBlock2(8, k, dil)
        load_seeded(rb, seed)
        x = rng_tensor(rng, (2, 8, 64)).requires_grad_(True)
        y = rb(x)
        sdo = {"r." + kk: v for kk, v in rb.state_dict().items()}
        close(O.resblock2_forward(sdo, "r", x, k, dil), y, what="resblock2 k%d" % k)
        r = rng_tensor(rng, y.shape)
        params = [p for _, p in rb.named_parameters()]
        gr = grads_of([y], [r], [x] + params)
        save("resblock2_k%d.npz" % k, seed=seed, dil=np.array(dil), x=x, y=y, r=r, dx=gr[0],
             **{"dp_" + n: gg for (n, _), gg in zip(rb.named_parameters(), gr[1:])})


if __name__ == "__main__":
    main()
