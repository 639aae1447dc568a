"""Registration demo with the reference's command line (/root/reference/src/demo.py:51-56):

    python demo.py --example {0..4} [--threshold T]

Loads the example pair, runs the MI355X-native RegTR forward (same batch / output contract as demo.py:176-190) and
prints the estimated pose and overlap statistics.  The reference opens a 4-pane VTK window (cvhelpers.visualization,
demo.py:59-139); that viewer is not part of this package -- pass --save OUT.npz to keep everything it would draw
(source / target clouds, key points, predicted correspondences, overlap scores, pose).
If the pretrained checkpoint is absent the model runs with random weights (as the reference's test.py does).
Extra flags: --data_dir / --ckpt_dir (default: the reference's ../data and ../trained_models), --save.
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_examples = [
    # 3DMatch examples (demo.py:26-38)
    ('3dmatch/ckpt/model-best.pth', 'indoor/test/7-scenes-redkitchen/cloud_bin_0.pth', 'indoor/test/7-scenes-redkitchen/cloud_bin_5.pth'),
    ('3dmatch/ckpt/model-best.pth', 'indoor/test/sun3d-hotel_umd-maryland_hotel3/cloud_bin_8.pth',
     'indoor/test/sun3d-hotel_umd-maryland_hotel3/cloud_bin_15.pth'),
    ('3dmatch/ckpt/model-best.pth', 'indoor/test/sun3d-home_at-home_at_scan1_2013_jan_1/cloud_bin_38.pth',
     'indoor/test/sun3d-home_at-home_at_scan1_2013_jan_1/cloud_bin_41.pth'),
    # ModelNet examples (demo.py:40-47)
    ('modelnet/ckpt/model-best.pth', 'modelnet_demo_data/modelnet_test_2_0.ply', 'modelnet_demo_data/modelnet_test_2_1.ply'),
    ('modelnet/ckpt/model-best.pth', 'modelnet_demo_data/modelnet_test_630_0.ply', 'modelnet_demo_data/modelnet_test_630_1.ply'),
]

parser = argparse.ArgumentParser()
parser.add_argument('--example', type=int, default=0, help=f'Example pair to run (between 0 and {len(_examples) - 1})')
parser.add_argument('--threshold', type=float, default=0.5, help='Overlap probability above which a keypoint counts as overlapping.')
parser.add_argument('--data_dir', type=str, default='../data')
parser.add_argument('--ckpt_dir', type=str, default='../trained_models')
parser.add_argument('--save', type=str, default=None, help='write the visualisation payload to this .npz')


def main():
    opt = parser.parse_args()
    from regtr_amd import RegTR, load_config
    from regtr_amd.harness import load_point_cloud
    ckpt_rel, src_rel, tgt_rel = _examples[opt.example]
    ckpt_path = os.path.join(opt.ckpt_dir, ckpt_rel)
    cfg_path = Path(ckpt_path).parents[1] / 'config.yaml'                       # demo.py:159
    if not cfg_path.exists():
        cfg_path = Path(ROOT) / 'regtr_amd' / 'conf' / ('3dmatch.yaml' if opt.example < 3 else 'modelnet.yaml')
    cfg = load_config(str(cfg_path))
    if not torch.cuda.is_available():
        sys.exit('regtr_amd runs on an MI355X (HIP) device only; there is no CPU path')
    device = torch.device('cuda:0')

    model = RegTR(cfg).to(device)
    if os.path.exists(ckpt_path):
        state = torch.load(ckpt_path, map_location=device, weights_only=False)
        model.load_state_dict(state['state_dict'])                              # strict, demo.py:165
    else:
        print(f'[demo] checkpoint {ckpt_path} not found: running with random weights')

    src_xyz = load_point_cloud(os.path.join(opt.data_dir, src_rel))
    tgt_xyz = load_point_cloud(os.path.join(opt.data_dir, tgt_rel))
    if 'crop_radius' in cfg:                                                     # demo.py:171-175
        src_xyz = src_xyz[np.linalg.norm(src_xyz, axis=1) < cfg['crop_radius'], :]
        tgt_xyz = tgt_xyz[np.linalg.norm(tgt_xyz, axis=1) < cfg['crop_radius'], :]

    data_batch = {'src_xyz': [torch.from_numpy(src_xyz).float().to(device)],
                  'tgt_xyz': [torch.from_numpy(tgt_xyz).float().to(device)]}
    outputs = model(data_batch)

    b = 0
    pose = outputs['pose'][-1, b].cpu().numpy()
    src_kp = outputs['src_kp'][b].cpu().numpy()
    src2tgt = outputs['src_kp_warped'][b][-1].cpu().numpy()                      # predicted positions of src_kp in the target
    overlap = torch.sigmoid(outputs['src_overlap'][b][-1]).cpu().numpy()
    m = overlap[:, 0] > opt.threshold
    np.set_printoptions(precision=6, suppress=True)
    print(f'source {len(src_xyz)} pts, target {len(tgt_xyz)} pts -> {len(src_kp)} source keypoints, '
          f'{int(m.sum())} predicted inside the overlap (p > {opt.threshold})')
    print('estimated pose (source -> target):')
    print(pose)
    if opt.save:
        np.savez(opt.save, src_xyz=src_xyz, tgt_xyz=tgt_xyz, src_kp=src_kp, src2tgt=src2tgt, src_overlap=overlap, pose=pose,
                 src_registered=src_xyz @ pose[:, :3].T + pose[:, 3])
        print(f'saved {opt.save}')


if __name__ == '__main__':
    main()
