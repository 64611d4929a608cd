#!/usr/bin/env python3
"""bench.py - hypotheses/sec through the D3DP DDIM loop on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one full D3DP.forward: T=10 DDIM steps with flip-TTA over one synthetic batch of B=1 H3WB clip
(27 frames x 134 keypoints) and P=20 hypotheses per GPU (BASELINE configs[2]: "H3WB paper setting, P=20 T=10").
Inputs, weights and noise are resident in HBM before the timed region.  With N > 1 the hypothesis axis is
sharded (weak scaling: P = 20*N, 20 per rank) and every step ends with the single RCCL all-gather of the
per-rank predictions.  Rank 0 prints ONE JSON line.

`value` is the default product mode, 'bf16x3': every GEMM operand carries all 24 bits of the fp32 number it stands for (three
bf16 slices, six bf16 MFMA products, fp32 accumulation).  The same run (N = 1) also times the opt-in `f16x2` mode and reports it
as a named secondary object, never as `value`: `value_f16x2` (three fp16 MFMA products on 22 - 23-bit operands: faster, narrower
than the reference's arithmetic).

Extra objects in the line:
  roofline      the dominant kernel family, the linear-layer launches of one flip-TTA denoiser pass - bf16x3: `sgemm2_kernel`
                (qkv, fc1: the persistent strip kernel of round 6), `gemm_dma_kernel` (proj, fc2), or the shared grids of the single-stream schedule
                with --streams 0; bf16x3_images: `xfqa_kernel` (qkv + attention), `xgemm_kernel`; f16x2: `hfqa_kernel`,
                `hgemm_kernel`, `hmlp_kernel`; f32: `gemm_kernel`.  Algorithmic FLOPs (2 M N K per linear layer) of those
                launches divided by their HIP-event time, launched back to back on the stream torch uses, against 416.7
                TFLOP/s (bf16 matrix peak / 6 products), 833.3 (fp16 / 3) or the 157.3 TFLOP/s f32 matrix peak.  The timed
                loop runs the three parts on three streams (queues), where a kernel's own duration is not observable (kernels
                of different queues share the CUs); the replay runs the SAME kernels of the SAME launches one after the
                other, which is the kernel-quality number, and `roofline_loop` is what the overlap makes of it.
                `by_layer`: each of the four layer kinds replayed alone (pafuse_d3dp_replay_layers), so the line shows
                which kernel of the family sits where (qkv / proj+LN / fc1+GELU / fc2+LN).
  roofline_loop the same fraction for the whole timed loop (2*T*69.38 GFLOP per hypothesis, everything included).
                `frac_timed_loop` inside it is `roofline_loop.frac`: the same fraction for the whole timed three-stream loop.
  cpu_baseline  the CPU oracle (a port of the reference's ATen path, oracle/) timed on the host cores of this box on the
                metric's own configuration (B=1, P=20, T=10 - unscaled).
  parity        the HIP default run on the very inputs, weights and noise of the cpu_baseline sample, against the oracle's output:
                max |d| of the poses and |dMPJPE| per protocol (max over the T steps), with the count of (step, protocol) pairs inside
                north_star's 1e-4 mm (BASELINE's metric is "hypotheses/sec ...; MPJPE mm"; consumer: main_h3wb.py:344-362).

N > 1: `python bench.py --gpus N` started WITHOUT torch.distributed's environment starts the N ranks itself - a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, before this process touches the GPU - relays
rank 0's line and exits with the children's status; started under torch.distributed.run it is one of the ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_DTYPE = "bf16x3"   # split-precision products on operands that carry all 24 bits of the fp32 numbers (three bf16 slices each):
#                            pafuse_amd.D3DP's default
GFLOP_PER_HYP_PASS = 69.384706048          # SURVEY.md section 2b / BASELINE.md section 3 (one denoiser pass)
PEAK_F32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: dense f32-input matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0             # dense bf16 matrix peak (opt-in --dtype bf16 runs are priced against this)
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6   # bf16x3: six bf16 MFMA products per fp32-equivalent product = 416.7
PEAK_F16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3   # f16x2: three fp16 MFMA products per fp32-equivalent product = 833.3
SPLIT_DTYPES = ("bf16x3", "bf16x3_images", "f16x2")   # the split-precision modes
SECONDARY_DTYPES = ("f16x2",)                          # timed in the same run as the default: a value_<mode> object, never `value`
#                                                        (round 6: 'bf16x3_images' - slower than the default, no wider - is no longer a bench leg)
_F32_NOTE = f"; the same FLOPs against the f32-input matrix peak {PEAK_F32_MFMA_TFLOPS}: frac_of_f32_peak"
MODES = {   # ceiling of the arithmetic scheme (TFLOP/s of fp32-equivalent work), matrix instructions per useful product, texts
    "f32": {"peak": PEAK_F32_MFMA_TFLOPS, "products": 1, "mfma": "v_mfma_f32_32x32x2_f32", "family": "gemm_kernel",
            "label": "f32 (fp32-input matrix cores: a k-ordered fp32 FMA chain per output)", "peak_note": "dense f32-input matrix peak"},
    "bf16x3": {"peak": PEAK_SPLIT_TFLOPS, "products": 6, "mfma": "qkv, fc1: v_mfma_f32_16x16x32_bf16; proj, fc2: v_mfma_f32_32x32x16_bf16",
               "family": "sgemm2_kernel (qkv, fc1), gemm_dma_kernel (proj, fc2)",
               "label": "bf16x3 (every fp32 GEMM operand as the exact sum of three bf16 slices, six bf16 MFMA products per fp32-equivalent "
                        "product, fp32 accumulate; activations, LayerNorm, softmax, attention and everything in memory fp32; the residual "
                        "stream stored centred on its row means)",
               "peak_note": "dense bf16 matrix peak 2500 / 6 products" + _F32_NOTE},
    "bf16x3_images": {"peak": PEAK_SPLIT_TFLOPS, "products": 6, "mfma": "v_mfma_f32_32x32x16_bf16",
                      "family": "xfqa_kernel (whose attention phase is inside the timed launches), xgemm_kernel",
                      "label": "bf16x3 on the image pipeline (the six-product scheme with both GEMM operands pre-split by their producers: "
                               "activations between kernels live as three-bf16-slice images, 6 bytes per element, exact; qkv + attention "
                               "fused; LayerNorm, softmax and attention arithmetic fp32)",
                      "peak_note": "dense bf16 matrix peak 2500 / 6 products" + _F32_NOTE},
    "f16x2": {"peak": PEAK_F16X2_TFLOPS, "products": 3, "mfma": "v_mfma_f32_32x32x16_f16; the fused qkv projection v_mfma_f32_16x16x32_f16",
              "family": "hgemm_kernel, hfqa_kernel (whose attention phase is inside the timed launches), hmlp_kernel",
              "label": "f16x2 (OPT-IN, narrower than the reference's fp32 operands: activations as two fp16 slices = 22 - 23 significant "
                       "bits, weights as two stored fp16 slices of the power-of-two-scaled tensor + one derived, three fp16 MFMA products "
                       "per product, fp32 accumulate; the residual stream, the attention output and the MLP hidden live in memory only "
                       "as two-slice images; LayerNorm, softmax and attention arithmetic fp32)",
              "peak_note": "dense fp16 matrix peak 2500 / 3 products" + _F32_NOTE},
    "bf16": {"peak": PEAK_BF16_MFMA_TFLOPS, "products": 1, "mfma": "v_mfma_f32_32x32x16_bf16", "family": "gemm_kernel",
             "label": "bf16 operands, f32 accumulate (opt-in, not the parity path)", "peak_note": "dense bf16 matrix peak"},
}


def spawn_ranks(n):
    """`python bench.py --gpus N` without torch.distributed's environment: start the N ranks as a child torch.distributed.run (a
    fresh process tree - this process has not touched the GPU and never will), relay its output, exit with its status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--proposals", type=int, default=20, help="hypotheses per GPU")
    ap.add_argument("--timesteps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="clips per forward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the value_f16x2 leg of the default line")
    ap.add_argument("--streams", type=int, default=2, help="aux HIP streams the three parts are spread over")
    ap.add_argument("--graph", action="store_true", help="replay the loop as one captured hipGraph")
    ap.add_argument("--fuse-qkv-attention", choices=("auto", "on", "off"), default="auto",
                    help="A/B: the fused qkv + attention kernel (pafuse_amd.MixSTE2.fuse_qkv_attention) where it exists; auto = "
                         "the modules' default: on in f16x2, always on in bf16x3_images, off in bf16x3")
    ap.add_argument("--fuse-mlp", choices=("auto", "on", "off"), default="auto",
                    help="A/B (f16x2): fc1 -> GELU -> fc2 of a block in one kernel (pafuse_amd.MixSTE2.fuse_mlp); auto = on where it exists")
    ap.add_argument("--fuse-mlp-parts", default="body,face,hands", help="A/B: the parts --fuse-mlp on applies to")
    ap.add_argument("--f32-residual", action="store_true",
                    help="A/B (image pipelines): keep the residual stream between the blocks as fp32 rows beside its image "
                         "(pafuse_amd.MixSTE2.keep_f32_residual)")
    ap.add_argument("--no-ln-fold", action="store_true",
                    help="A/B: the whole-row kernels write the normalised rows instead of folding norm1 / norm2 into the "
                         "qkv / fc1 GEMMs (pafuse_amd.MixSTE2.fold_layernorm)")
    ap.add_argument("--dtype", choices=("f32", "bf16x3", "bf16x3_images", "f16x2", "bf16"), default=DEFAULT_DTYPE,
                    help="matrix-product mode of the linear layers.  bf16x3 (default) = split precision at the reference's operand "
                         "width: fp32 operands as three bf16 slices (exact), six bf16 MFMA products, fp32 accumulate; bf16x3_images = "
                         "the same products on round 5's image pipeline (operands pre-split by their producers, qkv + attention fused); "
                         "f32 = fp32-input matrix cores; f16x2 = OPT-IN split precision on the fp16 matrix cores (activations as two "
                         "fp16 slices = 22 - 23 bits, three MFMA products): faster, NOT the reference's operand width, its own parity "
                         "bounds; bf16 = opt-in reduced precision (operands rounded to one bf16) - never the contract's line")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend for N > 1: nccl (= RCCL over xGMI, the contract's line); gloo only for "
                         "rehearsing the N > 1 code path on a box with fewer GPUs than ranks (with --single-device)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (never a performance number; needs --backend gloo)")
    ap.add_argument("--dump-output", default=None, metavar="FILE",
                    help="rank 0 saves the last step's gathered predictions [B,T,P,F,J,3] (torch.save): the rehearsal tests "
                         "compare them with a single-process run of the same seed")
    ap.add_argument("--train-dtype", choices=("f32", "bf16x3"), default="bf16x3",
                    help="--train: matrix products of the plain GEMMs (qkv, fc1, dX): split precision (default) or fp32 MFMA")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the `train` object of the inference line (a short B=37 training measurement, N = 1 only)")
    ap.add_argument("--dump-grads", default=None, metavar="FILE",
                    help="--train rehearsal: ONE forward + backward (no optimiser step, DropPath off, per-sample draws a function of the "
                         "GLOBAL sample index), rank 0 saves {parameter name: gradient} (torch.save) - under DDP the gradients averaged "
                         "over the ranks - and exits: tests compare them with a single-process step on the concatenated batch")
    ap.add_argument("--train", action="store_true",
                    help="time training steps instead (SURVEY 8f n2: fwd + bwd + AdamW, DDP over RCCL for N > 1); "
                         "--batch is then clips per GPU (default 37 = 1024 // 27, main_h3wb.py:781)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)            # (does not return; nothing above has initialised the GPU)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from pafuse_amd import synthetic as gu

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but torch.distributed's environment says WORLD_SIZE={world}")
    if args.single_device:
        if args.backend != "gloo":
            raise SystemExit("--single-device shares one GPU between the ranks: RCCL cannot, use --backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where the small collective operands live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    ge.build()
    if args.train:
        return train_bench(args, rank, local_rank, world, dev)
    from pafuse_amd import _lib
    from pafuse_amd.parallel import ShardedSampler, gather_hypotheses, rank_census, shard_range
    import ctypes as C

    B, T, P_local = args.batch, args.timesteps, args.proposals
    P_total = P_local * world
    x2d, x2f = gu.synthetic_inputs_2d(B=B)
    x2d, x2f = x2d.to(dev), x2f.to(dev)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def make(dtype):
        model, sd = ge.make_model(P_total, T, seed=51, device=dev)
        model.n_aux_streams = args.streams
        model.use_graph = args.graph
        model.precision = dtype
        for part_name, m in model.denoisers().items():
            if args.no_ln_fold:
                m.fold_layernorm = False
            m.fuse_qkv_attention = {"auto": None, "on": True, "off": False}[args.fuse_qkv_attention]
            m.keep_f32_residual = bool(args.f32_residual)
            m.fuse_mlp = {"auto": None, "on": part_name in args.fuse_mlp_parts.split(","), "off": False}[args.fuse_mlp]
        return model, sd

    def timed(model, steps, warmup):
        """W untimed + exactly K timed forward calls between fences -> (seconds per step, last output)"""
        sampler = ShardedSampler(model)
        torch.manual_seed(1234)              # identical on every rank: each draws the full-P noise, keeps its slice
        for _ in range(warmup):
            out = sampler(x2d, None, input_2d_flip=x2f)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = sampler(x2d, None, input_2d_flip=x2f)
        fence()
        elapsed = time.perf_counter() - t0
        assert out.shape == (B, T, P_total, 27, 134, 3) and bool(torch.isfinite(out).all())
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed / steps, out

    def family_replay(model, dtype, lanes, by_layer=True):
        """the linear-layer launches of one flip-TTA denoiser pass replayed back to back on torch's stream, HIP-event timed"""
        lib = _lib.load()
        mode = MODES[dtype]
        cfg = model.config_struct(True)
        nbytes = lib.pafuse_d3dp_workspace_bytes(C.byref(cfg), B, P_local)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev)
        reps = 3
        # the replay launches what the timed loop launched: part by part when the loop ran on side streams, the shared
        # grids of the single-stream bf16x3 schedule otherwise (pafuse_d3dp_config.part_by_part_launches, per call)
        per_part = lanes > 1 or dtype != "bf16x3"
        cfg.part_by_part_launches = int(per_part)

        def replay(mask):
            fl = C.c_double(0.0)
            nl = _lib.check(lib.pafuse_d3dp_replay_layers(C.byref(cfg), B, P_local, ws.data_ptr(), nbytes, stream.cuda_stream, mask, C.byref(fl)))  # warm-up
            fl = C.c_double(0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                _lib.check(lib.pafuse_d3dp_replay_layers(C.byref(cfg), B, P_local, ws.data_ptr(), nbytes, stream.cuda_stream, mask, C.byref(fl)))
            e1.record(stream)
            torch.cuda.synchronize(dev)
            return nl, fl.value, e0.elapsed_time(e1)

        launches, flops, ms = replay(15)
        n = launches * reps
        achieved = flops / (ms * 1e-3) / 1e12
        obj = {"bound": "mfma", "kernel": f"pafuse linear-layer GEMM family: {mode['family'] if per_part or dtype != 'bf16x3' else 'sgemm2_kernel, grouped_rowln_kernel'} ({mode['mfma']})",
               "schedule": (f"timed loop: {lanes} streams, one body-part denoiser per stream; this object: the same launches replayed one "
                            "after the other on one stream" if lanes > 1 else "timed loop and this replay: one stream"),
               "achieved": round(achieved, 2), "peak": mode["peak"], "unit": "TFLOP/s", "frac": round(achieved / mode["peak"], 4),
               "peak_note": mode["peak_note"], "frac_of_f32_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
               # the same launches seen two more ways: the useful FLOPs / the executed matrix work against the dense peak of the
               # instruction that runs
               "frac_of_bf16_dense_peak": None if dtype == "f32" else round(achieved / PEAK_BF16_MFMA_TFLOPS, 4),
               "mfma_pipe_frac": round(achieved * mode["products"] / (PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS), 4),
               "mfma_pipe_note": "executed matrix FLOPs (useful x 6 products in bf16x3, x 3 in f16x2) / dense peak of the instruction",
               "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "flops_per_launch": round(flops / n / 1e9, 3),
               "flops_unit": "GFLOP (algorithmic 2*M*N*K)"}
        if by_layer:
            layer_kernel = {"f16x2": {1: "hfqa_kernel: qkv projection + attention in one kernel (every block of every part)",
                                      2: "hgemm_kernel<..EPI_ROWLN> (one launch per part)",
                                      4: "hmlp_kernel: fc1 + GELU + fc2 + residual + LayerNorms in one kernel (face, hands; its FLOPs are both "
                                         "layers'); hgemm_kernel<..EPI_BIAS> (body)",
                                      8: "hgemm_kernel<..EPI_ROWLN> (the body only: fc2 of the face and the hands is inside the fused MLP launch)"},
                            "bf16x3_images": {1: "xfqa_kernel: qkv projection + attention in one kernel (every block of every part)",
                                              2: "xgemm_kernel<..EPI_ROWLN>", 4: "xgemm_kernel<..EPI_BIAS> (+ GELU, image out)", 8: "xgemm_kernel<..EPI_ROWLN>"},
                            "bf16x3": ({1: "sgemm2_kernel (persistent strip kernel, one launch per part)", 2: "gemm_dma_kernel<..EPI_ROWLN> (one launch per part)",
                                        4: "sgemm2_kernel (+ GELU; one launch per part)", 8: "gemm_dma_kernel<..EPI_ROWLN> (one launch per part)"} if per_part else
                                       {1: "sgemm2_kernel (one launch per part)", 2: "grouped_rowln_kernel", 4: "sgemm2_kernel (one launch per part)", 8: "grouped_rowln_kernel"}),
                            }.get(dtype, {1: "gemm_kernel", 2: "gemm_kernel<..EPI_ROWLN>", 4: "gemm_kernel", 8: "gemm_kernel<..EPI_ROWLN>"})
            obj["by_layer"] = {}
            for bit, name in ((1, "qkv"), (2, "proj+LN"), (4, "fc1+GELU"), (8, "fc2+LN")):
                nl, fl, t = replay(bit)
                if nl == 0:     # fc2 lives inside the fused MLP kernel, which the fc1 replay runs
                    obj["by_layer"][name] = {"kernel": "(inside the fused MLP kernel: see fc1+GELU)", "launches": 0}
                    continue
                obj["by_layer"][name] = {"kernel": layer_kernel[bit], "launches": nl * reps, "avg_launch_us": round(t * 1e3 / (nl * reps), 2),
                                         "achieved": round(fl / (t * 1e-3) / 1e12, 2), "frac": round(fl / (t * 1e-3) / 1e12 / mode["peak"], 4)}
            obj["by_layer_note"] = "each layer kind of the pass replayed alone (same tiles, HIP events); achieved in TFLOP/s, frac against `peak`"
        return obj, launches, ms / reps

    model, sd = make(args.dtype)
    sec_per_step, out = timed(model, args.steps, args.warmup)
    # what the collective layer saw (one all-gather of (rank, local hypothesis count)) and the all-gather's own cost
    lo, hi = shard_range(P_total, rank, world)
    census = rank_census(hi - lo)
    if census["ranks_seen"] != world or sum(census["P_local"]) != P_total:
        raise SystemExit(f"rank census {census} does not match --gpus {world} x {P_local} hypotheses")
    gather_ms = copy_ms = None
    if args.dump_output and rank == 0:
        torch.save(out.cpu(), args.dump_output)
    if world > 1:
        local = out[:, :, lo:hi].contiguous()
        gather_hypotheses(local, P_total)                      # warm-up
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            gather_hypotheses(local, P_total)
        fence()
        gt = torch.tensor([(time.perf_counter() - t0) / 5], dtype=torch.float64, device=cdev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_ms = round(float(gt.item()) * 1e3, 3)
        # of which the one pass that moves the rank axis next to the hypothesis axis (no communication)
        stage = torch.empty((world,) + tuple(local.shape), device=local.device if args.backend == "nccl" else "cpu")
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            stage.permute(1, 2, 0, 3, 4, 5, 6).reshape(out.shape)
        fence()
        copy_ms = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    value = B * P_total / sec_per_step
    # HIP streams (hardware queues) the library spread one rank's loop over
    lanes = _lib.check(_lib.load().pafuse_d3dp_lanes(C.byref(model.config_struct(True)), B, P_local, args.streams))
    flop_per_step = B * P_total * 2 * T * GFLOP_PER_HYP_PASS / 1e3 / world      # TFLOP per GPU and step

    def loop_roofline(dtype, sec):
        tf, peak = flop_per_step / sec, MODES[dtype]["peak"]
        return {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                "frac_of_f32_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "note": "whole timed loop per GPU: B*P*2*T*69.3847 GFLOP / step time"}

    folded = args.dtype in SPLIT_DTYPES and not args.no_ln_fold
    line = {
        "metric": "hypotheses/sec through DDIM loop (H3WB 133-kp, P=20, T=10)",
        "value": round(value, 3), "unit": "hypotheses/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": MODES[args.dtype]["label"],
        "data": "synthetic",
        "config": {"workload": f"D3DP.forward flip-TTA DDIM loop, H3WB 27x134 clips, B={B}, P={P_local}/GPU "
                               f"(P={P_total} total), T={T}, part-based MixSTE2 body/face/hands 384/224/256 ch, depth 8",
                   "B": B, "P_per_gpu": P_local, "P_total": P_total, "T": T, "flip_tta": True, "precision": args.dtype,
                   "parallelism": f"hypothesis-sharded x{world} + 1 all-gather" if world > 1 else "single GPU",
                   "collective_backend": (("nccl (RCCL)" if args.backend == "nccl" else "gloo (REHEARSAL, not a "
                                           "performance number)") if world > 1 else None),
                   "single_device_rehearsal": bool(args.single_device),
                   "weights": "seeded synthetic (no checkpoint offline)", "noise": "torch.randn on device (Philox)"},
        "kernel_source_sha256": _lib.kernel_source_digest(),
        "streams": lanes, "layernorm_folded_into_gemms": bool(folded),
        "residual_stream_in_memory": ("fp32 rows" if not folded or args.f32_residual or args.dtype == "bf16x3" else
                                      "H image only (two fp16 slices, 22-23 significant bits)" if args.dtype == "f16x2" else
                                      "X image only (three bf16 slices: the fp32 number exactly)") + (", centred on the row means" if folded else ""),
        "qkv_attention_fused_blocks": {name: _lib.check(_lib.load().pafuse_mixste2_fused_blocks(C.byref(m.weights_struct())))
                                       for name, m in model.denoisers().items()} if args.dtype in SPLIT_DTYPES else None,
        "ranks_seen": census["ranks_seen"], "P_local_per_rank": census["P_local"],
        "allgather_ms": gather_ms, "allgather_ms_note": "collective + the one layout pass, max over ranks", "gather_copy_ms": copy_ms,
        "roofline_loop": loop_roofline(args.dtype, sec_per_step),
    }

    # ---- dominant kernel family: the linear-layer GEMM launches of one flip-TTA denoiser pass, HIP-event timed ----
    if not args.no_roofline and rank == 0:
        obj, launches, pass_ms = family_replay(model, args.dtype, lanes)
        n, ms = obj["launches"], obj["avg_launch_us"] * obj["launches"] / 1e3
        # HBM bytes per launch come from rocprofv3 PMC passes of this same command (rocprof cannot run inside the benchmark):
        # the committed summary is quoted only when it was taken on THIS tree's kernel sources and in this mode.
        traffic, traffic_info = None, {"traffic_source": None}
        tpath = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
        if os.path.exists(tpath) and (B, P_local, T) == (1, 20, 10):
            tj = json.load(open(tpath))
            if tj.get("dtype") != args.dtype:
                traffic_info = {"traffic_source": f"profiles/r06_pmc_traffic.json was collected in {tj.get('dtype')} mode: not quoted"}
            elif tj.get("kernel_source_sha256") == _lib.kernel_source_digest():
                traffic = round(tj["traffic_bytes_per_launch"])
                traffic_info = {"traffic_source": "profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                                  "FETCH doubled per the gfx950 correction)",
                                "hbm_GBps_dominant_kernel": round(traffic / (ms * 1e-3 / n) / 1e9, 1),
                                "hbm_GBps_by_kernel_family": tj.get("hbm_GBps_by_kernel_family")}
            else:
                traffic_info = {"traffic_source": "profiles/r06_pmc_traffic.json is from other kernel sources "
                                                  "(kernel_source_sha256 differs): not quoted"}
        # algorithmic bytes per launch, two ways: (i) this design's launch boundaries in the fp32-activation modes (every GEMM
        # reads A and W and writes its outputs; whole-row kernels also read the residual and write x; per token and block, in
        # floats of C: qkv reads 1 writes 3; attention reads 3 writes 1; proj reads o, x writes x; fc1 reads 1 writes 2; fc2
        # reads 2, x writes x = 14 with the LayerNorm folded (8 B of statistics per row instead of xn), 16 without; + the 139.8 MB
        # of weights once); (ii) SURVEY 8(d)'s fused figure - weights once per pass + one read and one write of the residual stream
        # per block: 140 MB + 2*16*sum(M*C)*4 B per flip-TTA pass - both over the pass' launches
        rows = 2 * B * P_local * 27
        mc = rows * (24 * 384 + 68 * 224 + 42 * 256)
        per_token = 14 if folded else 16
        alg_unfused = (mc * 4 * per_token * 16 + 139.8e6) / launches
        alg_fused = (139.8e6 + 2 * 16 * mc * 4) / launches
        obj.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "algorithmic_bytes_per_launch": {"this_design_launch_boundaries_fp32_activations": round(alg_unfused),
                                                     "survey_8d_fused_blocks": round(alg_fused)},
                    "traffic_over_algorithmic": None if traffic is None else {
                        "vs_this_design": round(traffic / alg_unfused, 2), "vs_survey_8d": round(traffic / alg_fused, 2)},
                    **traffic_info})
        obj["frac_timed_loop"] = line["roofline_loop"]["frac"]
        obj["frac_timed_loop_note"] = ("roofline_loop.frac: the whole timed loop (three streams, attention and everything else included) against the "
                                       "same peak; `frac` is this kernel family replayed alone")
        line["roofline"] = obj

    # ---- the opt-in product modes, timed in the same run (N = 1, default invocation only): named objects, never `value` ----
    if world == 1 and args.dtype == DEFAULT_DTYPE and not args.no_secondary:
        del model
        for dtype in SECONDARY_DTYPES:
            m2, _ = make(dtype)
            sec2, _ = timed(m2, args.steps, max(args.warmup, 1))
            lanes2 = _lib.check(_lib.load().pafuse_d3dp_lanes(C.byref(m2.config_struct(True)), B, P_local, args.streams))
            sub = {"value": round(B * P_total / sec2, 3), "unit": "hypotheses/s", "ms_per_step": round(sec2 * 1e3, 3), "steps": args.steps,
                   "dtype": MODES[dtype]["label"], "roofline_loop": loop_roofline(dtype, sec2),
                   "note": "opt-in mode, timed in this run after the default; NOT the headline" +
                           ": its operands are narrower than the reference's fp32 (22 - 23 bits)"}
            if not args.no_roofline:
                sub["roofline"], _, _ = family_replay(m2, dtype, lanes2, by_layer=False)
            line[f"value_{dtype}"] = sub
            del m2

    # ---- CPU baseline: the oracle on the host cores, on the metric's own configuration; parity of the HIP default on that very sample
    if not args.no_cpu_baseline and rank == 0:
        from oracle import d3dp_oracle as orc
        Pc, Tc = 20, T        # the metric's P and T (unscaled since round 6: ~ 50 s of host CPU at the best thread count)
        noises = gu.synthetic_noises(B=1, P=Pc, n=Tc, seed=9)
        xc, xcf = gu.synthetic_inputs_2d(B=1)
        n1 = gu.synthetic_noises(B=1, P=1, n=1, seed=9)

        def one():
            t0 = time.perf_counter()
            orc.ddim_sample(sd, xc, n1, 1, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=xcf)
            return time.perf_counter() - t0

        # ATen's CPU GEMMs do not scale to every hardware thread of a big host: pick the best thread count on a
        # one-hypothesis, one-step probe (first call per setting is a warm-up), then time the sample with it.
        hw = os.cpu_count() or 1
        probe = {}
        for n in sorted({min(hw, c) for c in (8, 16, 32, 64)}):   # (all 256 threads of a big host: minutes per call)
            torch.set_num_threads(n)
            one()
            probe[n] = one()
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        ref = orc.ddim_sample(sd, xc, noises, Tc, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=xcf)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(Pc / dt, 4), "unit": "hypotheses/s", "cores": cores,
                                "kind": "port",
                                "sample": f"oracle/d3dp_oracle.py (torch CPU fp32) flip-TTA loop B=1 P={Pc} T={Tc} (the metric's configuration, "
                                          f"unscaled) in {dt:.2f} s; thread count picked from a P=1,T=1 probe {{threads: s}} = "
                                          f"{ {k: round(v, 2) for k, v in probe.items()} } on a {hw}-thread host"}
        # the HIP default on the same inputs, weights and noise (the oracle here is the checker, never the thing measured)
        if args.dtype == DEFAULT_DTYPE and world == 1 and (B, P_local) == (1, Pc):
            mp, _ = make(args.dtype)
            mp.noise_fn = lambda k, shape, device: noises[k].to(device)
            out = mp(xc.to(dev), None, input_2d_flip=xcf.to(dev)).cpu()
            del mp
            line["parity"] = parity_object(orc, out, ref, gu, xc, Pc, Tc)

    # ---- BASELINE configs[4] beside the headline (N = 1): a short training measurement, so that the driver's record holds one
    if world == 1 and not args.no_train_leg:
        line["train"] = train_leg(dev, args.streams)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()      # ranks > 0 wait here while rank 0 runs the roofline and CPU-baseline legs
        dist.destroy_process_group()


def parity_object(orc, out, ref, gu, x2d, P, T):
    """|HIP - oracle| on one sample: pointwise and on the four MPJPE protocols of main_h3wb.py:327-362 (fp64 metric arithmetic on
    whole-body poses against the seeded synthetic target; J-Agg on the joints where both runs pick the same hypothesis)."""
    import torch
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    cam = torch.tensor([[2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001]], dtype=torch.float64)
    traj = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64)

    def whole(pred):
        pw = orc.wb_pose_from_parts(pred.double())
        Bq, Tq, Pq, Fq = pw.shape[:4]
        rp = orc.project_to_2d((pw + traj).reshape(-1, 134, 3), cam.repeat(Bq * Tq * Pq * Fq, 1)).reshape(Bq, Tq, Pq, Fq, 134, 2)
        return pw, rp

    tw = orc.wb_pose_from_parts(target.double())
    x2 = x2d.double()
    (pa, ra), (pb, rb) = whole(out), whole(ref)
    d = {"J-Best": (orc.j_best(pa, tw) - orc.j_best(pb, tw)).abs() * 1000,
         "P-Best": (orc.p_best(pa, tw) - orc.p_best(pb, tw)).abs() * 1000,
         "P-Agg": (orc.p_agg(pa, tw) - orc.p_agg(pb, tw)).abs() * 1000}
    # J-Agg: per (step, frame, joint) the hypothesis with the smallest 2-D reprojection error; compared where both runs pick the same one
    e2a, e2b = (ra - x2[:, None, None]).norm(dim=-1), (rb - x2[:, None, None]).norm(dim=-1)
    e3a, e3b = (pa - tw[:, None, None]).norm(dim=-1), (pb - tw[:, None, None]).norm(dim=-1)
    ia, ib = e2a.argmin(dim=2), e2b.argmin(dim=2)
    same = ia == ib
    n = same.sum(dim=(0, 2, 3)).clamp(min=1)
    d["J-Agg"] = ((((e3a.gather(2, ia[:, :, None]).squeeze(2) - e3b.gather(2, ib[:, :, None]).squeeze(2)) * same).sum(dim=(0, 2, 3)) / n).abs() * 1000)
    met = sum(int((v <= 1e-4).sum()) for v in d.values())
    tot = sum(v.numel() for v in d.values())
    return {"vs": "oracle fp32 (oracle/d3dp_oracle.py: pinned to the reference bit for bit on G1-G19)", "P": P, "T": T,
            "max_abs_m": float((out - ref).abs().max()), "mean_abs_m": float((out - ref).abs().mean()),
            "dMPJPE_mm": {k: float(v.max()) for k, v in d.items()},
            "dMPJPE_mm_mean_over_steps": {k: float(v.mean()) for k, v in d.items()},
            "pairs_within_1e-4_mm": f"{met} of {tot}",
            "j_agg_different_picks": float((~same).double().mean()),
            "note": "|MPJPE_hip - MPJPE_oracle| per protocol, max over the T steps; north_star asks 1e-4 mm, two correct fp32 "
                    "implementations of this network differ by 2 - 4e-4 mm (DESIGN.md section 4)"}


def train_leg(dev, streams, B=37, steps=6):
    """The `train` object of the inference line: D3DP train step (fwd + bwd + AdamW) at B = 37 clips, one GPU, 'bf16x3' products
    (the single-process default; `python bench.py --train` is the full training bench with its CPU baseline and DDP)."""
    import torch
    import __graft_entry__ as ge
    from pafuse_amd import synthetic as gu
    model, _ = ge.make_model(1, 1, seed=51, device=dev, is_train=True)
    model.n_aux_streams = streams
    # (the side streams are the inference model's: pafuse_amd.d3dp keeps one set per device and process - a second set would
    # share the process' hardware queues with the first: 376 clips/s with five streams alive against 435 - 441 with three)
    x2d, _ = gu.synthetic_inputs_2d(B=B, seed=1234)
    target = gu.synthetic_target_3d(B=B, seed=1235).to(dev)
    x2d = x2d.to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=6e-5, weight_decay=0.1)
    torch.manual_seed(4321)

    def step():
        opt.zero_grad(set_to_none=True)
        pred = model(x2d, target)
        loss = torch.mean(torch.norm(pred - target, dim=-1))
        loss.backward()
        opt.step()
        return loss

    for _ in range(6):      # the optimiser allocates its state in the first step, the caching allocator settles in the second, and
        step()              # the chip has idled through the CPU baseline: a few steps bring its clocks back up
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(loss.detach()))
    tflops = B * 3 * GFLOP_PER_HYP_PASS / 1e3 / sec
    return {"metric": "training clips/sec (H3WB 27x134 clips, fwd+bwd+AdamW), BASELINE configs[4] on one GPU", "value": round(B / sec, 2),
            "unit": "clips/s", "ms_per_step": round(sec * 1e3, 2), "steps": steps, "B": B, "dtype": model.precision,
            "achieved": round(tflops, 2), "unit_achieved": "TFLOP/s of fp32-equivalent work (B*3*69.3847 GFLOP / step)",
            "frac": round(tflops / MODES[model.precision]["peak"], 4), "peak": MODES[model.precision]["peak"],
            "frac_note": "against the ceiling of the step's product scheme (bf16x3: 2500 / 6), as the inference line is quoted",
            "frac_of_f32_peak": round(tflops / PEAK_F32_MFMA_TFLOPS, 4)}


def train_bench(args, rank, local_rank, world, dev):
    """One training step = D3DP.forward in train mode (per-sample q_sample targets, three per-part denoisers with
    DropPath), the caller's mpjpe loss, backward through the HIP kernels, AdamW (lr 6e-5, wd 0.1, main_h3wb.py:761).
    N > 1: torch DistributedDataParallel - one RCCL all-reduce of the 35 M fp32 gradients, bucketed and overlapped
    with the backward pass.  Weak scaling: `--batch` clips per GPU."""
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from pafuse_amd import synthetic as gu

    B = args.batch if args.batch > 1 else 37
    model, _ = ge.make_model(1, 1, seed=51, device=dev, is_train=True)
    model.n_aux_streams = args.streams
    train_dtype = args.train_dtype     # 'bf16x3' (default): split products in qkv / fc1 / every dX GEMM; 'f32': fp32 MFMA everywhere
    model.precision = train_dtype
    effective = model.prepare_for_ddp()      # N > 1: the precision in effect beside the collective's kernels (pafuse_amd.D3DP.prepare_for_ddp); reported below
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) if world > 1 else model
    x2d, _ = gu.synthetic_inputs_2d(B=B, seed=1234 + rank)
    target = gu.synthetic_target_3d(B=B, seed=1235 + rank).to(dev)
    x2d = x2d.to(dev)
    if args.dump_grads:
        # the two-rank rehearsal of the DDP step (tests/test_hip_train.py): everything random is a function of the global sample index
        for m in model.denoisers().values():
            m.drop_path_rate = 0.0

        def draw(i, base=rank * B):
            g = torch.Generator().manual_seed(7000 + base + i)
            return torch.randint(0, 1000, (1,), generator=g), torch.randn(27, 134, 3, generator=g)
        model.train_draw_fn = draw
        pred = net(x2d, target)
        torch.mean(torch.norm(pred - target, dim=-1)).backward()
        torch.cuda.synchronize(dev)
        if rank == 0:
            torch.save({"precision": effective, "world": world, "B_per_rank": B,
                        "grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}}, args.dump_grads)
            print(json.dumps({"metric": "training gradient dump (rehearsal)", "n_gpus": world, "dtype": effective, "file": args.dump_grads}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    opt = torch.optim.AdamW(net.parameters(), lr=6e-5, weight_decay=0.1)
    torch.manual_seed(4321 + rank)

    def step():
        opt.zero_grad(set_to_none=True)
        pred = net(x2d, target)
        loss = torch.mean(torch.norm(pred - target, dim=-1))          # common/loss.py:27-34 (mpjpe)
        loss.backward()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss.detach()))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    sec = elapsed / args.steps
    tflops = B * 3 * GFLOP_PER_HYP_PASS / 1e3 / sec                  # per GPU: forward + 2x backward
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the oracle's train step (torch CPU autograd over the functional restatement) on one clip
        from oracle import d3dp_oracle as orc
        model.train_draw_fn = None
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        leaves = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in sd.items()}
        xc, tc = x2d[:1].cpu(), target[:1].cpu()
        g = torch.Generator().manual_seed(3)
        tt, nz = torch.tensor([500]), torch.randn(1, 27, 134, 3, generator=g)
        torch.set_num_threads(min(os.cpu_count() or 1, 16))

        def cpu_step():
            t0 = time.perf_counter()
            pred = orc.train_forward(leaves, xc, orc.q_sample_targets(sd, tc, tt, nz), tt)
            orc.mpjpe(pred, tc).backward()
            return time.perf_counter() - t0

        cpu_step()
        dt = min(cpu_step() for _ in range(2))
        cpu = {"value": round(1.0 / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle/d3dp_oracle.py train_forward + mpjpe + torch autograd backward, 1 clip, {dt:.2f} s "
                         f"(no optimiser step)"}
    if rank == 0:
        print(json.dumps({
            "metric": "training clips/sec (H3WB 27x134 clips, fwd+bwd+AdamW)", "value": round(B * world / sec, 3),
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": effective, "dtype_requested": train_dtype, "data": "synthetic",
            "config": {"workload": f"D3DP train step, B={B} clips/GPU, per-part MixSTE2 384/224/256 ch depth 8, "
                                   f"DropPath 0.1, AdamW", "B_per_gpu": B,
                       "parallelism": f"DDP x{world} (RCCL all-reduce of 35 M fp32 grads)" if world > 1 else "single GPU",
                       "weights": "seeded synthetic", "peak_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)},
            "roofline_loop": {"bound": "mfma", "achieved": round(tflops, 2), "peak": MODES[effective]["peak"],
                              "unit": "TFLOP/s", "frac": round(tflops / MODES[effective]["peak"], 4),
                              "frac_of_f32_peak": round(tflops / PEAK_F32_MFMA_TFLOPS, 4),
                              "note": "per GPU: B*3*69.3847 GFLOP (forward + 2x backward) / step time"},
            "cpu_baseline": cpu}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
