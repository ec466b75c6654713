"""CPU checks of the stage-1 and stage-2 oracles (test infrastructure) against what pins them:
* oracle/mapping_ref.py vs the golden sequences produced by the reference's own Semantic_Mapping (bit for bit);
* oracle/rcnn_ref.py vs the known-answer vectors of detectron2's own published unit tests (v0.6:
  tests/layers/test_roi_align.py::test_forward_output, tests/modeling/test_anchor_generator.py::
  test_default_anchor_generator) -- the only numeric fixtures that exist for this stage, detectron2 itself being
  neither vendored nor installed -- plus definitional properties of the remaining operators."""
import os

import numpy as np
import pytest
import torch

from oracle import mapping_ref, rcnn_ref


@pytest.mark.parametrize("fname,ncat", [("mapping_golden.npz", 10), ("mapping_golden_c22.npz", 22)])
def test_mapping_oracle_reproduces_reference_golden(golden_dir, fname, ncat):
    z = np.load(os.path.join(golden_dir, fname))
    cfg = mapping_ref.MapCfg(num_sem_categories=ncat)
    for name in sorted({k.split("/")[0] for k in z.files}):
        depth, sem, rel = z[f"{name}/depth"], z[f"{name}/sem"], z[f"{name}/pose_obs"]
        maps = torch.zeros(4 + ncat, cfg.map_cells, cfg.map_cells)
        pose = torch.tensor([cfg.local_size_cm / 100.0 / 2.0, cfg.local_size_cm / 100.0 / 2.0, 0.0])
        for i in range(min(depth.shape[0], 4)):                       # four frames per sequence keep the suite fast
            obs = np.zeros((1, 4 + ncat) + depth[i].shape, np.float32)
            obs[0, 3] = depth[i]
            obs[0, 4:] = sem[i].astype(np.float32)
            with torch.no_grad():
                fp, maps, _, pose = mapping_ref.forward(torch.from_numpy(obs), torch.from_numpy(rel[i]), maps, pose, cfg)
            assert np.array_equal(np.packbits(fp.numpy().astype(bool)), z[f"{name}/fp_map_bits"][i])
            assert np.array_equal(pose.numpy(), z[f"{name}/poses"][i])
            assert np.array_equal(maps.double().sum((1, 2)).numpy(), z[f"{name}/channel_sums"][i])
            assert np.array_equal((maps != 0).sum((1, 2)).numpy(), z[f"{name}/channel_nnz"][i])


def test_roi_align_matches_detectron2_known_answers():
    """detectron2 tests/layers/test_roi_align.py::test_forward_output: 5x5 arange map, box (1,1,3,3), 4x4 output."""
    feat = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    rois = torch.tensor([[0.0, 1.0, 1.0, 3.0, 3.0]])
    old = [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]]            # aligned=False
    new = [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]]  # aligned=True
    assert torch.allclose(rcnn_ref.roi_align(feat, rois, 1.0, 4, sampling_ratio=0, aligned=False)[0, 0], torch.tensor(old))
    assert torch.allclose(rcnn_ref.roi_align(feat, rois, 1.0, 4, sampling_ratio=0, aligned=True)[0, 0], torch.tensor(new))
    # same test file, test_resize: a 2x2 box of a map equals the box of the 2x down-scaled map at scale 0.5
    g = torch.Generator().manual_seed(0)
    big = torch.rand((1, 1, 10, 10), generator=g)
    small = torch.nn.functional.avg_pool2d(big, 2)
    r = torch.tensor([[0.0, 2.0, 2.0, 8.0, 8.0]])
    a = rcnn_ref.roi_align(small, r, 0.5, 3, sampling_ratio=0, aligned=True)
    b = rcnn_ref.roi_align(small, torch.tensor([[0.0, 1.0, 1.0, 4.0, 4.0]]), 1.0, 3, sampling_ratio=0, aligned=True)
    assert torch.allclose(a, b)


def test_anchors_match_detectron2_known_answers():
    """detectron2 tests/modeling/test_anchor_generator.py::test_default_anchor_generator: sizes (32, 64), aspect ratios
    (0.25, 1, 4), a 1x2 feature map of stride 4 (anchors ordered location-major, then size, then ratio)."""
    expected = torch.tensor([[-32.0, -8.0, 32.0, 8.0], [-16.0, -16.0, 16.0, 16.0], [-8.0, -32.0, 8.0, 32.0],
                             [-64.0, -16.0, 64.0, 16.0], [-32.0, -32.0, 32.0, 32.0], [-16.0, -64.0, 16.0, 64.0],
                             [-28.0, -8.0, 36.0, 8.0], [-12.0, -16.0, 20.0, 16.0], [-4.0, -32.0, 12.0, 32.0],
                             [-60.0, -16.0, 68.0, 16.0], [-28.0, -32.0, 36.0, 32.0], [-12.0, -64.0, 20.0, 64.0]])
    ratios = (0.25, 1.0, 4.0)
    a32 = rcnn_ref.grid_anchors((1, 2), 4, 32, ratios).view(2, 3, 4)
    a64 = rcnn_ref.grid_anchors((1, 2), 4, 64, ratios).view(2, 3, 4)
    got = torch.cat([a32, a64], dim=1).reshape(-1, 4)
    assert torch.allclose(got, expected)


def test_box_transform_nms_and_paste_definitions():
    g = torch.Generator().manual_seed(1)
    boxes = torch.rand((20, 2), generator=g) * 50
    boxes = torch.cat([boxes, boxes + torch.rand((20, 2), generator=g) * 30 + 2], 1)
    # zero deltas are the identity; a delta built from a target box reproduces it (Box2BoxTransform round trip)
    w = (10.0, 10.0, 5.0, 5.0)
    assert torch.allclose(rcnn_ref.apply_deltas(torch.zeros(20, 4), boxes, w), boxes, atol=1e-5)
    tgt = boxes + torch.tensor([1.0, -2.0, 3.0, 0.5])
    bw, bh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    d = torch.stack([w[0] * ((tgt[:, 0] + 0.5 * tw) - (boxes[:, 0] + 0.5 * bw)) / bw,
                     w[1] * ((tgt[:, 1] + 0.5 * th) - (boxes[:, 1] + 0.5 * bh)) / bh,
                     w[2] * torch.log(tw / bw), w[3] * torch.log(th / bh)], 1)
    assert torch.allclose(rcnn_ref.apply_deltas(d, boxes, w), tgt, atol=1e-4)
    # greedy NMS by its definition (brute force over the sorted list)
    cats = torch.randint(0, 2, (20,), generator=g)
    keep = rcnn_ref.nms_sorted(boxes, cats, 0.3)
    kept = []
    for i in range(20):
        ok = True
        for j in kept:
            if cats[i] != cats[j]:
                continue
            ix = max(0.0, min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]))
            iy = max(0.0, min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]))
            inter = float(ix * iy)
            union = float((boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1]) + (boxes[j, 2] - boxes[j, 0]) * (boxes[j, 3] - boxes[j, 1])) - inter
            if inter / union > 0.3:
                ok = False
                break
        if ok:
            kept.append(i)
    assert keep.nonzero().flatten().tolist() == kept
    # a mask of ones pasted into its box fills exactly the pixels whose centres lie inside the box
    m = rcnn_ref.paste_masks(torch.ones((1, 28, 28)), torch.tensor([[3.0, 2.0, 9.0, 7.0]]), (10, 12), 0.5)[0]
    want = torch.zeros((10, 12), dtype=torch.bool)
    want[2:7, 3:9] = True
    assert torch.equal(m, want)


def test_nms_iou_matches_published_known_answers():
    """The IoU behind rcnn_ref.nms_sorted against detectron2's tests/structures/test_boxes.py::TestBoxIOU::
    test_pairwise_iou (unit box vs six boxes: 1, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)), probed through the
    suppression decision on either side of each value; and torchvision's test/test_ops.py::TestNMS::test_nms_float16
    boxes (three near-duplicates, scores 0.6370 / 0.7569 / 0.3966, threshold 0.2: only the top-scoring box survives)."""
    unit = [0.0, 0.0, 1.0, 1.0]
    others = [[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0],
              [0.5, 0.5, 1.5, 1.5]]
    expected = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]
    for box, iou in zip(others, expected):
        pair = torch.tensor([unit, box])
        cats = torch.zeros(2, dtype=torch.int64)
        assert rcnn_ref.nms_sorted(pair, cats, iou - 1e-4).tolist() == [True, False], (box, iou)     # IoU > thr: suppressed
        assert rcnn_ref.nms_sorted(pair, cats, iou + 1e-4).tolist() == [True, True], (box, iou)      # IoU <= thr: kept
        assert rcnn_ref.nms_sorted(pair, torch.tensor([0, 1]), 0.0).tolist() == [True, True]         # other category: never
    boxes = torch.tensor([[285.3538, 185.5758, 1193.5110, 851.4551], [285.1472, 188.7374, 1192.4984, 851.0669],
                          [279.2440, 197.9812, 1189.4746, 849.2019]])
    scores = torch.tensor([0.6370, 0.7569, 0.3966])
    keep = rcnn_ref.batched_nms(boxes, scores, torch.zeros(3, dtype=torch.int64), 0.2)
    assert keep.tolist() == [1]


def test_vectorised_roi_align_equals_the_loop_form():
    """oracle/rcnn_ref.roi_align_vec (used where 1000 proposals per image make the loop form impractical) against
    roi_align -- itself pinned on detectron2's known answers above -- on boxes inside, across and outside the map,
    sub-pixel boxes, both alignment conventions, adaptive and fixed sampling: equal to the bit."""
    import torch
    from oracle import rcnn_ref
    g = torch.Generator().manual_seed(0)
    feat = torch.randn((2, 8, 25, 34), generator=g)
    rois = torch.tensor([[0, 1.2, 3.4, 60.7, 55.1], [1, -5.0, -3.0, 20.0, 30.0], [0, 100.0, 80.0, 140.0, 100.5], [1, 10, 10, 10.5, 10.2],
                         [0, 0, 0, 135.9, 99.9], [1, 130, 90, 150, 120], [0, 33.3, 7.7, 34.1, 90.0]])
    for scale, P, sr, aligned in ((0.25, 7, 0, True), (0.25, 14, 0, True), (0.125, 7, 2, False), (0.25, 4, 0, False), (1.0, 3, 0, True)):
        assert torch.equal(rcnn_ref.roi_align(feat, rois, scale, P, sr, aligned), rcnn_ref.roi_align_vec(feat, rois, scale, P, sr, aligned, chunk=3))
    pyr = {k: torch.randn((2, 4, 64 >> i, 80 >> i), generator=g) for i, k in enumerate(("p2", "p3", "p4", "p5"))}
    r = torch.tensor([[0, 4.0, 4.0, 40.0, 60.0], [1, 0.0, 0.0, 300.0, 250.0], [0, 10.0, 20.0, 130.0, 140.0], [1, 50.0, 50.0, 52.0, 51.0]])
    assert torch.equal(rcnn_ref.roi_pool(pyr, r, 7), rcnn_ref.roi_pool(pyr, r, 7, vectorised=True))


def test_pil_resize_restatement_is_pinned_on_pillow(golden_dir):
    """detectron2 resizes uint8 frames with PIL (ResizeTransform.apply_image -> Image.resize(BILINEAR)); Pillow is not
    part of the reference checkout, so oracle/rcnn_ref.pil_resize_bilinear_u8 restates its published two-pass fixed-point
    algorithm.  Pinned on outputs of Pillow itself: the committed vectors (oracle/gen_golden_resize.py) and, where Pillow
    is importable, a live comparison at the agent's geometry (480x640 -> 800x1067) and a downscale."""
    import numpy as np
    from oracle import rcnn_ref
    z = np.load(os.path.join(golden_dir, "pil_resize_golden.npz"))
    i = 0
    while f"in{i}" in z:
        want = z[f"out{i}"]
        assert np.array_equal(rcnn_ref.pil_resize_bilinear_u8(z[f"in{i}"], want.shape[0], want.shape[1]), want), i
        i += 1
    assert i >= 5
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.RandomState(7)
    for (h, w, nh, nw) in ((480, 640, 800, 1067), (600, 900, 400, 600)):
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(rcnn_ref.pil_resize_bilinear_u8(img, nh, nw), np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)))
