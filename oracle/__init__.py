"""CPU oracle for the sam_road tiled-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``sam_road_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and there only as the checker.

What it is
----------
A plain-PyTorch, eager, fp32 restatement of the reference's algorithm for the
path ``inferencer.py -> SAMRoad.infer_masks_and_img_features / infer_toponet /
SAMRoad.forward`` (reference ``model.py:414-508``), using the reference's own
``state_dict`` key names so one state dict drives both the
oracle and the HIP path.

PARITY UNPINNED (by the reference's own tests)
----------------------------------------------
* The reference ships no test that pins any numeric output of the model
  (SURVEY.md §4), and the SAM ViT encoder arithmetic lives in an un-vendored,
  un-pinned git submodule (``.gitmodules:1-6`` -> ``htcr/segment-anything-road``,
  a fork of ``facebookresearch/segment-anything``; commit unknown, directory
  empty in the snapshot).  ``model.py`` itself cannot be imported here
  (lightning / torchmetrics / wandb / torchvision / the fork are absent).
* The oracle is therefore anchored on what *is* available:
    - the encoder restatement is cross-checked against the independent
      ``transformers.models.sam.modeling_sam.SamVisionEncoder`` (installed,
      eager attention) — ``tests/test_oracle_golden.py`` (``test_encoder_matches_hf_golden``,
      ``test_encoder_matches_hf_at_true_vitb_512_dims``);
    - ``BilinearSampler`` / ``TopoNet`` / ``get_patch_info_one_img`` /
      ``nms_points`` are checked against golden vectors produced by executing
      the reference's *own source* for those definitions (extracted by AST from
      ``/root/reference/model.py``, ``dataset.py``, ``graph_utils.py`` at fixture
      generation time) — ``tests/golden/make_golden.py`` (committed) and
      ``tests/test_oracle_golden.py``;
    - torch behaviours the path relies on are kept as known-answer tests
      (SURVEY.md §8c).

Each function cites the reference ``file:line`` it follows.
"""
