import sys, torch
sys.path.insert(0, 'image-matching-webui_amd'); sys.path.insert(0, '.')
from imcui_hip import backend
from imcui_hip.pipeline import SuperPointLightGluePipeline
from imcui_hip.synth import make_pair_batch
from oracle.weights import lightglue_state_dict, superpoint_state_dict
dev = torch.device('cuda', 0)
B = int(sys.argv[1])
pipe = SuperPointLightGluePipeline(
    {"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
    {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)},
).eval().to(dev)
img0, img1, _ = make_pair_batch(1234, B, 480, 640, distinct=min(B, 4))
img0, img1 = img0.to(dev), img1.to(dev)
f = pipe.extractor.forward_batched(torch.cat([img0, img1], 0))
torch.cuda.synchronize(); print('SP ok', f['keypoints'].shape, flush=True)
k0, k1 = f["keypoints"][:B], f["keypoints"][B:]
d0, d1 = f["descriptors"][:B], f["descriptors"][B:]
n0, n1 = f["num_keypoints"][:B], f["num_keypoints"][B:]
m = pipe.matcher.forward_batched(k0, k1, d0, d1, n0, n1, (640, 480), (640, 480))
torch.cuda.synchronize(); print('LG ok', flush=True)
for key in ("matches0", "matching_scores0"):
    v = m[key]
    bad = []
    for i in range(4, B):
        if not torch.equal(v[i], v[i % 4]):
            bad.append((i, int((v[i] != v[i % 4]).sum())))
    print(key, 'replica mismatches:', bad, 'nan:', bool(torch.isnan(v.float()).any()), flush=True)
