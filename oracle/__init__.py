"""ORACLE - CPU restatement of the reference's hot-path algorithms.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package ``millieye_amd`` never imports it.

  darknet_ref.py   detector forward (stock torch CPU ops)        pinned by tests/golden/darknet_*.npz
  network_ref.py   Network.forward (m3) heads / assembly / losses pinned by tests/golden/network_*.npz
  tv_ops.c/.py     torchvision roi_align / ps_roi_align / nms     PARITY UNPINNED (dependency absent,
                                                                  no reference fixture; see tv_ops.c)
  import_reference.py  build-container-only recipe that imports /root/reference to make the fixtures
"""
