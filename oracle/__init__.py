"""oracle/ -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (torch fp32 / numpy) of the reference's text+vision fusion
train-step hot path.  It exists so that the
HIP path can be checked against the reference's arithmetic on a machine where
/root/reference does not exist (the GPU box).

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- and there only as the checker / the timed
CPU baseline, never as the thing shipped.  Nothing under `ultrafnd_git_amd/`
imports it; the product fails loudly when its HIP library is missing.

Pinning (how the oracle itself is trusted):
  * Tier A (fusion + classifier + CE + clip + AdamW, metrics): PINNED.  The
    reference's own modules were imported in the build container
    (`tests/golden/make_golden.py`, transformers masked per SURVEY 8c) and the
    restatement was compared element-for-element (forward, every grad, params
    after 1 and 3 optimizer steps); the committed fixtures under
    `tests/golden/` hold the reference's outputs.
  * Tier B (BERT-base text encoder, CLIP ViT-B/32 visual encoder): the
    arithmetic is third-party (`transformers`, requirement `>=4.40.0`,
    unpinned in the reference's requirements.txt:7; 5.15.0 installed here).
    The restatement is compared against the locally installed model classes
    built from local configs with seeded random weights.  The reference's own
    tests hold no vectors at this boundary: "parity unpinned" beyond that
    third-party comparison (pretrained weights are not available offline).
"""
