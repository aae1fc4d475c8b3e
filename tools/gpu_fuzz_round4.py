"""Development tool (device): random shapes through the kernels added in round 4 -- the chained cLN (many more tiles than the chip holds
workgroups: the ticket order must keep the look-back chain free of deadlock), gLN over tokens in both forms, the attention core with and
without dropout -- each against torch's float64 autograd and the emulator, through the kernel tests' own checks.
    python tools/gpu_fuzz_round4.py [seconds]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_kernels as GK

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(20260927)
t0, n = time.time(), 0
big = [("cln", (24, 64, 20000)), ("cln", (3, 512, 9000)), ("cln", (130, 16, 3000)), ("prelu_cln", (40, 96, 6000, 0.2)), ("cln", (2, 300, 33))]
while time.time() - t0 < budget:
    if big:
        kind, args = big.pop(0)
    else:
        kind = rng.choice(["cln", "prelu_cln", "gln_tokens", "attn", "attn"])
        if kind == "cln":
            args = (rng.randint(1, 6), rng.choice([4, 24, 64, 100, 128, 200, 256, 300, 512]), rng.randint(1, 5000))
        elif kind == "prelu_cln":
            args = (rng.randint(1, 4), rng.choice([8, 48, 64, 128, 384, 512]), rng.randint(1, 3000), rng.choice([0.25, -0.3, 0.0, 1.5]))
        elif kind == "gln_tokens":
            args = (rng.randint(1, 200), rng.randint(1, 3000), rng.choice([4, 16, 64, 128, 256, 1024]))
            if args[0] * args[1] * args[2] > 3e7:
                continue
        else:
            args = (rng.randint(1, 40), rng.randint(1, 320), rng.randint(1, 8), rng.choice([8, 16, 32]), rng.choice([0.0, 0.0, 0.1, 0.5]))
            if args[4] > 0 and args[0] * args[1] * args[1] * args[2] < 20000:      # (the test checks the kept fraction: needs a sample of some size)
                args = args[:4] + (0.0,)
    fn = {"cln": GK.test_cln_fwd_bwd, "prelu_cln": GK.test_prelu_cln_fwd_bwd, "gln_tokens": GK.test_gln_tokens_fwd_bwd, "attn": GK.test_attention_core_fwd_bwd}[kind]
    try:
        fn(*args)
    except Exception as e:
        print("FAILED", kind, args, repr(e)[:300])
        sys.exit(1)
    n += 1
    if n <= 6 or n % 20 == 0:
        print("case {:3d} {} {} ok".format(n, kind, args), flush=True)
print("{} cases in {:.0f} s, all within the kernel tests' tolerances".format(n, time.time() - t0))
