"""bench.py -- images/sec of the VQ-VAE-2 stage-1 train step (BASELINE.json metric) on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + MSE+0.25*latent loss + backward + EMA codebook update + Adam on one batch of
synthetic 256x256 images already resident in HBM (BASELINE.json configs[1]: default two-level
VQVAE, batch 32 per GPU; weak scaling: every rank keeps 32 images, one packed RCCL all-reduce per
step).  Rank 0 prints ONE JSON line.  The line also carries
  roofline     -- the dominant kernel (fp32-MFMA 3x3 conv, csrc/vq2_wino.hip / vq2_conv.hip): algorithmic
                  FLOPs of its launches / their HIP-event time, measured live in the timed region
  cpu_baseline -- the CPU oracle (PyTorch-CPU restatement of the reference graph, kind "port") timed on
                  this box's host cores on a bounded sample (N=1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
WORKLOADS = {
    # name: (image size, per-GPU batch, n_embed, description)
    "c2": (256, 32, 512, "configs[1]: 256x256 FFHQ-shaped synthetic, default two-level VQ-VAE-2, batch 32/GPU"),
    "c4": (256, 32, 8192, "configs[3]: 256x256, n_embed=8192 large codebook, batch 32/GPU"),
    "c5": (512, 8, 512, "configs[4]: 512x512 synthetic, default two-level VQ-VAE-2, batch 8/GPU"),
    "tiny": (64, 4, 512, "debug: 64x64, batch 4"),
    # SURVEY 8f-4 (not a BASELINE config): the default VQVAE_Deep of vqvae_deep.py:234-261 -- channel 256, ResBlock(256,128)
    # x6 per stack, embed_dim 256, AdaIN decoder with style_dim 2048 -- through the drop-in module + one-launch Adam
    "deep": (256, 32, 512, "next-row f4: 256x256, default VQVAE_Deep (26.6 M parameters, style_dim 2048), batch 32/GPU"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # >= 100 timed steps after >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-all", action="store_true",
                    help="bracket EVERY GEMM launch (per-shape table; perturbs the step by ~6%%); default: only "
                         "the dominant kernel")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel table (JSON) here")
    return ap.parse_args()


def prof_report(lib):
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib.vq2_prof_report(buf, len(buf))
    assert rc == 0, lib.vq2_last_error()
    rows = {}
    for line in buf.value.decode().splitlines():
        name, n, ms, flops, nbytes = line.split()
        rows[name] = {"launches": int(n), "ms": float(ms), "flops": float(flops), "bytes": float(nbytes)}
    return rows


def cpu_baseline(size, n_embed, budget_s=7.0):
    """Oracle train step (same graph, same synthetic data) on the host cores, at 8 threads (the survey's
    figure), 32 threads and every core torch would use by default; `value` is the
    FASTEST of them (the stated baseline is the CPU's best, not an oversubscribed run)."""
    from oracle import vqvae_oracle as O
    cfg = O.VQVAEConfig(n_embed=n_embed)
    b = 8
    img = O.make_images(b, size, 1234)
    all_threads = torch.get_num_threads()
    runs = []
    for nt in sorted({min(8, all_threads), min(32, all_threads), all_threads}):
        torch.set_num_threads(nt)
        st = O.make_state(cfg, 1234)
        adam = O.AdamState({k: v for k, v in st.items() if not O.is_buffer(k)})
        O.train_step(st, cfg, img, adam)  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < budget_s and n < 40):
            O.train_step(st, cfg, img, adam)
            n += 1
        dt = time.perf_counter() - t0
        runs.append({"threads": nt, "value": round(b * n / dt, 3), "steps": n, "seconds": round(dt, 1)})
    torch.set_num_threads(all_threads)
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "images/s", "cores": best["threads"], "kind": "port",
            "sample": f"oracle train steps of batch {b} at {size}x{size}, 1 warm-up + {best['steps']} timed "
                      f"({best['seconds']} s) at {best['threads']} threads; torch {torch.__version__} CPU kernels",
            "runs": runs}


class DeepStep:
    """The reference's loop body (train_vqvae.py:83-91) on the drop-in VQVAE_Deep: forward(input, style), MSE + 0.25 *
    latent, backward, Adam -- the optimizer as ONE vq2_adam_step launch over a flat arena (weight gradients land in
    their arena slots straight from the wgrad kernels)."""
    dp = False

    def __init__(self, amd, model, style):
        from vqvae2_amd.optim import FusedAdam, ParamArena
        live = model.live_parameters()
        self.amd, self.model, self.style = amd, model, style
        self.arena = ParamArena(live)
        self.opt = FusedAdam(live, lr=3e-4, arena=self.arena)

    def step(self, img):
        self.arena.zero_grad()
        dec, diff, _ = self.model(img, style=self.style)
        loss, recon, latent = self.amd.stage1_loss(dec, diff, img)
        loss.backward()
        self.opt.step()
        return {"loss": loss.detach(), "recon": recon, "latent": latent}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    import vqvae2_amd
    # device binding + RCCL group ("nccl" is RCCL on ROCm) from the launcher environment; at world size 1 a group
    # exists only under VQ2_DP_FORCE=1 (exercises the collective path on one GPU)
    # (VQ2_BENCH_BACKEND=gloo + VQ2_SHARE_GPU=1: rehearsal of the N > 1 path with several ranks on ONE GPU, which RCCL
    #  refuses; used by tests/test_gpu_rccl.py -- the numbers of such a run mean nothing)
    backend = os.environ.get("VQ2_BENCH_BACKEND", "nccl")
    rank, local_rank, world = vqvae2_amd.distributed.bringup(backend)
    dev_index = 0 if os.environ.get("VQ2_SHARE_GPU", "0") != "0" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from oracle import vqvae_oracle as O
    lib = vqvae2_amd._lib.lib

    size, batch, n_embed, desc = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    cfg = O.VQVAEConfig(n_embed=n_embed)
    deep = args.workload == "deep"
    if deep:
        if world != 1:
            raise SystemExit("--workload deep is a single-GPU measurement of the drop-in VQVAE_Deep module")
        from oracle import vqvae_deep_oracle as OD
        args.no_cpu_baseline = True
        model = vqvae2_amd.VQVAE_Deep()
        model.load_state_dict(OD.make_deep_state(OD.DEEP_DEFAULT, 1234, 0.3, 1.5))
        model.to(dev).train()
        trainer = DeepStep(vqvae2_amd, model, OD.make_style(batch, OD.DEEP_DEFAULT, 1234).to(dev))
    else:
        model = vqvae2_amd.VQVAE(n_embed=n_embed)
        model.load_state_dict(O.make_state(cfg, 1234))  # identical replicas on every rank
        model.to(dev)
        trainer = vqvae2_amd.Stage1Trainer(model, lr=3e-4)
    img = O.make_images(batch, size, 1234, rank=rank).to(dev)  # resident in HBM before timing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # "recon-MSE parity" half of the BASELINE metric, recorded with the number: the first two images through the HIP
    # path and through the CPU oracle (eval forward: same weights, same codebooks), outside the timed region
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not deep:
        import torch.nn.functional as F
        st0 = O.make_state(cfg, 1234)
        model.eval()
        with torch.no_grad():
            dec_g, _ = model(img[:2])
            _, _, _, idt_g, idb_g = model.encode(img[:2])
            dec_c, _, idt_c, idb_c = O.vqvae_forward(st0, cfg, img[:2].cpu(), training=False)
        model.train()
        parity = {"images": 2,
                  "recon_mse_hip": float(F.mse_loss(dec_g, img[:2])), "recon_mse_cpu": float(F.mse_loss(dec_c, img[:2].cpu())),
                  "dec_max_abs_diff": float((dec_g.cpu() - dec_c).abs().max()),
                  "index_mismatches": int((idt_g.cpu() != idt_c).sum() + (idb_g.cpu() != idb_c).sum()),
                  "indices": int(idt_c.numel() + idb_c.numel())}

    for _ in range(args.warmup):
        out = trainer.step(img)
    barrier()
    # live roofline: the dominant kernel's launches are bracketed by HIP events on every 4th timed step (each pair of
    # events costs ~10 us of stream time: sampling keeps the measurement inside the timed region at a quarter of the cost)
    level = 0 if args.no_prof else (1 if (args.prof_all or args.kernel_table) else 2)
    every = 1 if level == 1 else 4
    sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = level and i % every == 0
        lib.vq2_prof_enable(level if on else 0)
        sampled += 1 if on else 0
        out = trainer.step(img)
    t_host = time.perf_counter() - t0   # host time to ENQUEUE the steps (launch-bound if close to dt)
    barrier()
    dt = time.perf_counter() - t0
    lib.vq2_prof_enable(0)
    kernels = prof_report(lib) if not args.no_prof else {}
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(out["loss"])
    if not (loss == loss):
        raise SystemExit("loss is NaN")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = batch * world * args.steps / dt
        roof = None
        fam = {}
        for k, v in kernels.items():   # fold the per-shape records into kernel families
            f = fam.setdefault(k.split("|")[0], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for kk in f:
                f[kk] += v[kk]
        conv = {k: v for k, v in fam.items() if k.startswith(("conv_gemm", "conv_wino3"))}
        if conv:
            dom = max(conv, key=lambda k: conv[k]["ms"])
            r = conv[dom]
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
            # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE doubled as the
            # gfx950 guide prescribes, + WRITE_SIZE); they cannot be collected inside this process.
            traffic, tsrc = None, None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pm = json.load(f).get(args.workload, {}).get(dom)
                if pm and pm.get("batch") == batch:
                    traffic, tsrc = pm["bytes_per_launch"], pm["source"]
            except OSError:
                pass
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": tsrc, "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"]),
                    "launches_per_step": r["launches"] // max(sampled, 1), "sampled_steps": sampled,
                    "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2)}
            if dom.startswith("conv_wino3"):
                # `achieved` counts the ALGORITHMIC FLOPs of the 3x3 convolution (18 per output and channel pair); the
                # Winograd F(2,3) form executes 2/3 of them on the matrix pipe (csrc/vq2_wino.hip), so the fraction of
                # the fp32 MFMA peak the hardware actually sustains is executed_frac
                roof["executed_tflops"] = round(ach * 2.0 / 3.0, 2)
                roof["executed_frac"] = round(ach * 2.0 / 3.0 / PEAK_F32_TFLOPS, 4)
                roof["note"] = ("achieved/frac count the ALGORITHMIC FLOPs of the 3x3 convolution (contract); the kernel runs the "
                                "F(2,3) minimal-filtering form in fp32 and executes 2/3 of them: executed_frac is the share of the "
                                "157.3 TFLOP/s fp32 MFMA peak the hardware sustains")
        line = {
            "metric": "images/sec VQVAE_Deep 256px train step (not the BASELINE metric)" if deep else
                      "images/sec VQ-VAE-2 256px train step", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "global_batch": batch * world, "image": size, "n_embed": n_embed,
                       "parallelism": f"dp{world}"},
            "final_loss": round(loss, 6), "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3),
            "roofline": roof,
        }
        if trainer.dp:
            # proof that the collective path saw `world` ranks: the communicator's own world size, the bytes of every
            # bucket it moved per step and how many of them left from a backward hook (overlapped with backward)
            line["collectives"] = {"backend": dist.get_backend(), "data_path": trainer.comm.name,
                                   "world": trainer.comm.world(), "buckets_per_step": len(trainer.bucket_bytes) or 1,
                                   "bucket_bytes": trainer.bucket_bytes or {"all": 4 * trainer.arena.flat_g.numel()},
                                   "early_buckets": trainer.early_buckets,
                                   "steps": args.steps + args.warmup}
        if parity is not None:
            line["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(size, n_embed)
        if kernels:
            line["kernel_ms_per_step"] = {k: round(v["ms"] / max(sampled, 1), 4) for k, v in fam.items()}
            if level == 1:   # every GEMM-shaped launch was bracketed: algorithmic FLOPs of the whole step
                gf = sum(v["flops"] for v in kernels.values()) / max(sampled, 1) / 1e9
                line["algorithmic_gflop_per_step"] = round(gf, 1)
                line["step_tflops"] = round(gf / ms, 2)
        if args.kernel_table:
            for v in kernels.values():
                v["tflops"] = round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)
                v["gbps"] = round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)      # algorithmic bytes / time
                v["us_per_launch"] = round(v["ms"] * 1e3 / v["launches"], 1)
            with open(args.kernel_table, "w") as f:
                json.dump(dict(sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])), f, indent=1)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
