"""The reference's per-frame loop (test/metrics_from_model.py:178-294) written against the drop-in mirrors, ONE FRAME PER
CALL, with the reference's own two timers: what a maintainer who applies INTEGRATION.md §2 to the reference sees.

Statement for statement the body of the reference loop (the same caller-side JSON work included: json.loads + json.dumps of
every camera's list, :183-191, and json.dumps of every person's skeletons, :253-257), `time_graph_matching` taken as at
:179/:229 and `time_3D` as at :236/:295.  The README of the reference quotes 31.67 ms (matching) and 19.65 ms (3D) per frame
for its own path on its authors' GPU; bench.py prints this loop's means beside them (`dropin_loop`).
"""
import json
import time

import numpy as np
import torch

README_MS = {'graph_matching': 31.67, 'pose_3d': 19.65}      # reference README.md ("Time" table), its authors' hardware


def build_models(gat_sd, prm, mlp_sd):
    from ..gat2 import GAT2 as GAT
    from ..mlp import PoseEstimatorMLP
    from ..parameters import parameters
    model = GAT(None, prm['gnn_layers'], prm['num_feats'], prm['n_classes'], prm['num_hidden'], prm['heads'],
                torch.nn.LeakyReLU(), torch.nn.Sigmoid(), prm['in_drop'], prm['attn_drop'], prm['alpha'], prm['residual'], bias=True)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in gat_sd.items()})
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # the reference's constructor prints its input size
        mlp = PoseEstimatorMLP(input_dimensions=len(parameters.cameras) * len(parameters.joint_list) * parameters.numbers_per_joint,
                               output_dimensions=len(parameters.joint_list) * 3)
    mlp.load_state_dict({k: torch.as_tensor(v) for k, v in mlp_sd.items()})
    return model, mlp


def run(frames, model, mlp, warmup=3, device=None):
    """frames: list of wire-format frame dicts.  -> dict(mean ms of both stages, frames/s, per-frame poses of the last frame)."""
    from ..graph_generator import MergedMultipleHumansDataset
    from ..parameters import parameters
    from ..pose_estimator_dataset_from_json import PoseEstimatorDataset
    from ..skeleton_matching_utils import get_person_proposal_from_network_output
    device = device or torch.device('cuda')
    CLASSIFICATION_THRESHOLD = 0.5
    t_gm = t_3d = 0.0
    n_gm = n_3d = n_frames = n_persons = 0
    last = None
    t_all0 = None
    inside = [0.0]                      # host time spent inside the package's symbols (the rest is the caller's own code)

    def call(fn, *a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        inside[0] += time.perf_counter() - t0
        return r
    for it, input_element in enumerate(frames):
        if it == warmup:
            torch.cuda.synchronize()
            t_gm = t_3d = 0.0
            n_gm = n_3d = n_frames = n_persons = 0
            inside[0] = 0.0
            t_all0 = time.time()
        time_ini = time.time()
        processed_input = dict()
        for cam in input_element:
            data = json.loads(input_element[cam][0])
            cam_data = []
            for s in data:
                cam_data.append(s)
            if cam_data:
                processed_input[cam] = []
                processed_input[cam].append(json.dumps(cam_data))
                processed_input[cam].append(input_element[cam][1])
        scenario = call(MergedMultipleHumansDataset, processed_input, mode='test', limit=10000, debug=True,
                        alt=parameters.graph_alternative, verbose=False)
        if len(scenario.graphs) == 0:
            continue
        subgraph = scenario.graphs[0].to(device)
        indices = scenario.data['edge_nodes_indices'][0].to(device)
        nodes_camera = scenario.data['nodes_camera'][0]
        feats = call(lambda: subgraph.ndata['h']).to(device)
        model.g = subgraph
        for layer in model.layers:
            layer.g = subgraph
        outputs = torch.squeeze(call(model, feats.float(), subgraph))
        indices = torch.squeeze(indices).to('cpu')
        final_output = call(get_person_proposal_from_network_output, outputs, subgraph, indices, nodes_camera, scenario.jsons_for_head,
                            CLASSIFICATION_THRESHOLD)
        time_GM_i = time.time() - time_ini
        if len(final_output) > 0:
            t_gm += time_GM_i
            n_gm += 1
        time_a = time.time()
        final_results = list()
        batched_input = []
        for person in final_output:
            raw_input = dict()
            for cam_idx, camera in enumerate(parameters.used_cameras):
                if person[camera] is not None:
                    pc = person[camera]
                    all_joints_data = [scenario.jsons_for_head[pc]]
                    raw_input[camera] = [json.dumps(all_joints_data)]
            inputs = call(PoseEstimatorDataset, raw_input, parameters.cameras, parameters.joint_list, save=False)
            if inputs.__len__() == 0:
                continue
            inputs = inputs[0][0].reshape([1, inputs[0][0].size()[0]]).to(device)
            batched_input.append(inputs)
        if batched_input:
            input_all = torch.cat(batched_input, dim=0)
            output_all = call(mlp, input_all.to(device))
            for person_id in range(output_all.shape[0]):
                results_3d = torch.squeeze(output_all[person_id]) * 10.
                results_3d = results_3d.to('cpu')
                x3D = results_3d[::3]
                y3D = results_3d[1::3]
                z3D = results_3d[2::3]
                person_result = list()
                for idx_joint in range(len(parameters.joint_list)):
                    person_result.append(np.array([x3D[idx_joint], y3D[idx_joint], z3D[idx_joint]]))
                final_results.append(person_result)
        time_3D_i = time.time() - time_a
        if len(final_results) > 0:
            t_3d += time_3D_i
            n_3d += 1
        n_frames += 1
        n_persons += len(final_results)
        last = final_results
    torch.cuda.synchronize()
    dt = time.time() - (t_all0 if t_all0 is not None else time.time())
    return {'frames': n_frames, 'persons_per_frame': n_persons / max(1, n_frames),
            'graph_matching_ms': 1e3 * t_gm / max(1, n_gm), 'pose_3d_ms': 1e3 * t_3d / max(1, n_3d),
            'ms_per_frame': 1e3 * dt / max(1, n_frames), 'frames_per_s': n_frames / dt if dt > 0 else 0.0,
            # host time inside the package's symbols per frame; the remainder is the reference caller's own Python (json.loads /
            # json.dumps of every camera list and every person, tensor reshapes, .to('cpu') waits for the MLP kernels, numpy boxing)
            'inside_mirrors_ms': 1e3 * inside[0] / max(1, n_frames),
            'reference_readme_ms': README_MS, 'last': last}
