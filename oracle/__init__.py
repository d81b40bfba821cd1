"""CPU oracle of the TorchOk hot path — TEST INFRASTRUCTURE ONLY.

Nothing under torchok_amd/ imports this package.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may, and only as the checker / reported baseline.

Parity status: the reference pins NO numeric outputs for backbones/necks/heads (shape tests only,
tests/additional_tests/models/**); its one known-answer test on the hot path (JointLoss,
tests/base_tests/losses/test_base_losses.py:33,48,75-77) is reproduced in tests/.  The restatement
here is additionally checked, in the build container, against the reference's OWN importable
files (heads, losses, registry) and against the reference's own resnet.py wiring constructed on
top of the restated timm subset (tests/golden/gen_golden.py -> tests/golden/*.npz).  The [timm 0.6.13]
block semantics themselves come from recall (timm is not installed and has no
source under /root/reference) => backbone numerics are "parity unpinned" by the reference.
"""
