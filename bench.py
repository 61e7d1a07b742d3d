"""Throughput benchmark of the MI355X inference path (contract: see the task's bench rules).

One step = one pass of the hot path over one resident batch of synthetic frames:
mpe_match_batch (featurise + GAT + clustering) followed by mpe_mlp3d_batch (MLP 3D), i.e.
BASELINE.json configs[1] ("Panoptic 5-view, 4-person; GATv2 match + MLP 3D, 1k-frame batch on
1 MI355X").  With --mode tri the 3D stage is the DLT triangulation path (configs[2]).

Inputs (packed 2D skeletons) are resident in HBM before the timed region.  With N > 1 every
rank processes its own 1k-frame shard (frames are independent) and the 3D
poses are all-gathered over RCCL each step; value = frames of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      fp32-MFMA GEMM kernel (k_linear_dma): algorithmic FLOPs of its launches / their
                summed duration, measured with HIP events on the launch stream inside the
                timed region; peak = 157.3 TFLOP/s fp32 matrix (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/oracle_np.py, a port of the reference's algorithm on
                torch-CPU) timed on rank 0 on a bounded sample of the same frames.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = '3d_multi_pose_estimator_amd'

PEAK_FP32_MFMA_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=1000, help='frames per rank and step')
    ap.add_argument('--persons', type=int, default=4)
    ap.add_argument('--mode', choices=['mlp', 'tri'], default='mlp')
    ap.add_argument('--cpu-sample', type=int, default=40, help='frames of the CPU baseline sample (0 = skip)')
    ap.add_argument('--fast-mlp', action='store_true', help='plain fp32 accumulation in the MLP GEMMs')
    ap.add_argument('--bf16-mlp', action='store_true',
                    help='reduced precision (NOT the parity path): bf16 MFMA for the MLP GEMMs, configs[4] style')
    ap.add_argument('--reduced', action='store_true',
                    help='configs[4] precision (NOT the parity path): bf16 MFMA for the GAT and MLP GEMMs, fp16 '
                         'feature rows in the attention stage')
    ap.add_argument('--preset', default='PANOPTIC', choices=['PANOPTIC', 'ARPLAB', 'RING23'],
                    help='camera rig; RING23 = the 23-view stress rig of BASELINE.json configs[4] (fp32 here)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)')
    ap.add_argument('--streams', type=int, default=1, choices=[1, 2],
                    help='2 = software pipeline across steps: matching of batch i+1 overlaps the 3D stage of batch i')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    n_gpus = world if distributed else 1
    device = torch.device('cuda', local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)

    syn = importlib.import_module(PKG + '.synthetic')
    cal = importlib.import_module(PKG + '.calibration')
    par = importlib.import_module(PKG + '.parameters')
    pipeline = importlib.import_module(PKG + '.pipeline')
    params = par.select(args.preset)
    calib = cal.Calibration(params, syn.ring_transform_manager(params) if args.preset == 'RING23' else None)
    V, J = len(params.used_cameras_skeleton_matching), len(params.joint_list)
    nf = 2 + V * J * 10
    # logit shift chosen so that nearly every pair clears the 0.5 threshold: the greedy
    # clustering then assigns every skeleton, i.e. `persons` people per frame reach the MLP.
    gat_sd = syn.gat_state_dict(7, nf, logit_gain=25.0, logit_shift=0.698 + 0.25)
    prm = syn.gat_params(nf)
    in_dim = len(params.cameras) * J * params.numbers_per_joint
    mlp_sd = syn.mlp_state_dict(11, in_dim)

    B = args.frames
    spec = syn.FrameSpec(persons=args.persons)
    # distinct frames per rank (frame index = global index)
    uniq = min(B, 250)
    frames = []
    for i in range(uniq):
        f, _ = syn.make_frame(calib, rank * B + i, spec)
        frames.append({c: [f[c][0], f[c][1]] for c in f})
    frames = [frames[i % uniq] for i in range(B)]

    eng = pipeline.Engine(params, calib, max_frames=B, max_persons_per_camera=args.persons, device=str(device))
    eng.load_gat(gat_sd, prm)
    eng.load_mlp(mlp_sd)
    if args.fast_mlp:
        eng.set_precision(False, False)
    if args.bf16_mlp:
        eng.set_precision(False, False, mlp_bf16=True)
    if args.reduced:
        eng.set_precision(False, False, mlp_bf16=True, gat_reduced=True)
    pb = eng.pack(frames)
    db = eng.to_device(pb)
    torch.cuda.synchronize(device)

    gather_buf = None
    s_match = torch.cuda.Stream(device) if args.streams == 2 else None
    s_3d = torch.cuda.Stream(device) if args.streams == 2 else None
    keep = []          # tensors produced on one stream and consumed on the other stay referenced

    def stage3d(persons, n_persons):
        if args.mode == 'mlp':
            return eng.mlp3d(db, persons, n_persons)[0]
        return eng.triangulate(db, persons, n_persons)[0]

    def step():
        if args.streams == 2:
            # the two stages use disjoint workspace, so matching of the next step may run
            # while the 3D stage of this one is still in flight
            with torch.cuda.stream(s_match):
                _, persons, n_persons = eng.match(db, want_scores=False)
                ev = torch.cuda.Event()
                ev.record(s_match)
            with torch.cuda.stream(s_3d):
                s_3d.wait_event(ev)
                poses = stage3d(persons, n_persons)
            keep.append((persons, n_persons, poses))
            if len(keep) > 4:
                keep.pop(0)
            if not distributed:
                return poses, n_persons
            torch.cuda.current_stream(device).wait_stream(s_3d)
        else:
            _, persons, n_persons = eng.match(db, want_scores=False)
            poses = stage3d(persons, n_persons)
        if distributed:
            nonlocal gather_buf
            if gather_buf is None:
                gather_buf = torch.empty((world * poses.shape[0],) + tuple(poses.shape[1:]), dtype=poses.dtype, device=device)
                step.np_buf = torch.empty((world * B,), dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(gather_buf, poses)
            dist.all_gather_into_tensor(step.np_buf, n_persons)
        return poses, n_persons

    step()                       # initialisation (workspace, LDS attributes, communicator): not a warmup step
    torch.cuda.synchronize(device)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    eng.profile(True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses, n_persons = step()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    persons_per_frame = float(n_persons.float().mean().item())
    total_frames = B * n_gpus * args.steps
    value = total_frames / elapsed

    out = {
        'metric': 'frames/sec (5-view Panoptic, 4 persons) at 1/2/4/8 GPUs; MPJPE vs ref',
        'value': value, 'unit': 'frames/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16 GEMMs / f16 attention rows' if args.reduced else ('bf16 (MLP) / f32' if args.bf16_mlp else 'f32'), 'data': 'synthetic',
        'config': {'workload': ('%s: %d-view x %d-person, GAT match + %s, %d-frame batch per GPU'
                                % ('c2 Panoptic' if args.preset == 'PANOPTIC' else args.preset, V, args.persons, 'MLP 3D' if args.mode == 'mlp' else 'DLT triangulation', B)),
                   'frames_per_step_per_gpu': B, 'heads_per_batch': pb.n_heads, 'edge_nodes_per_batch': pb.n_edge_nodes,
                   'persons_found_per_frame': persons_per_frame, 'parallelism': 'frame-shard x%d' % n_gpus,
                   'streams': args.streams,
                   'mlp_accumulate': 'bf16 mfma (reduced precision)' if (args.bf16_mlp or args.reduced) else ('f32' if args.fast_mlp else 'f32 mfma + f64 running sums'),
                   'weights': 'deterministic hash init (no checkpoint offline)'},
    }
    if rank == 0:
        gemm_s = prof['gemm_ms'] * 1e-3
        achieved = prof['gemm_flop'] / gemm_s / 1e12 if gemm_s > 0 else 0.0
        out['roofline'] = {
            'kernel': 'mpe::k_linear_dma (fp32 MFMA 16x16x4 GEMM, LDS-DMA staging, fused bias + LeakyReLU)', 'bound': 'mfma',
            'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': pmc_traffic(),
            'launches': prof['gemm_launches'], 'avg_launch_ms': prof['gemm_ms'] / max(1, prof['gemm_launches']),
            'flop_per_step': prof['gemm_flop'] / args.steps,
            'gemm_share_of_step': gemm_s / elapsed,
            'flop_definition': 'algorithmic 2*M*N*K of the launches (layer-0 edge-node rows de-duplicated)'
                               + ('; NOTE: reduced-precision run, bf16 launches are priced against the fp32 peak here' if (args.bf16_mlp or args.reduced) else ''),
        }
        # the CPU baseline is taken on rank 0 at N = 1 only
        out['cpu_baseline'] = None if distributed else cpu_baseline(args, frames, calib, gat_sd, prm, mlp_sd)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def pmc_traffic():
    """Fabric bytes per k_linear launch from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    2x FETCH correction); None when the file is absent.  PMC cannot be sampled from inside
    the process, so this is the offline measurement, not a live one."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as fh:
            return json.load(fh)['k_linear_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, frames, calib, gat_sd, prm, mlp_sd):
    """CPU port (the oracle) on a bounded sample of the same frames, rank 0 only."""
    if args.cpu_sample <= 0:
        return None
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    onp = importlib.import_module('oracle_np')
    n = min(args.cpu_sample, len(frames))
    onp.run_frame(frames[0], calib, gat_sd, prm, mlp_sd, mode=args.mode)       # warm up
    t0 = time.perf_counter()
    for i in range(n):
        onp.run_frame(frames[i], calib, gat_sd, prm, mlp_sd, mode=args.mode)
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d frames of the same batch, per-frame loop as in the reference (torch-CPU GEMMs)' % n}


if __name__ == '__main__':
    main()
