"""oracle/ -- CPU restatement of the CNMNet hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``cnmnet_amd``) imports this directory.  The
only allowed consumers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; there it acts as the checker / the timed
CPU baseline, never as the thing shipped.

Parity status: the reference repository (xxlong0/CNMNet) ships NO tests and NO
golden vectors, so parity is pinned by outputs of the reference itself, imported
in the authoring container by ``oracle/import_reference.py`` and frozen as small
fixtures under ``tests/golden/`` by ``tests/golden/make_golden.py``.

Modules
  ref_arrangement.py   stock torch-CPU ops laid out as the reference lays them
                       out (per-plane grid_sample loop, Conv2d/BatchNorm2d/ReLU/
                       Upsample stacks, Unfold-based depth->normal).  Used as
                       the oracle for the full nets and as the timed CPU baseline.
  closed_form.py       independent numpy restatement (explicit bilinear taps,
                       closed-form 3x3 solve) of plane sweep, depth->normal and
                       inverse warp, with D/H/W free.  Cross-checks the torch
                       restatement and is the oracle for D != 64.
  import_reference.py  imports /root/reference unmodified (stubs + patches);
                       exists only where /root/reference exists.
"""
