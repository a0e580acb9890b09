"""CPU oracle for the ReLaX-VQA feature-extraction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline, never as the thing shipped.  The product path (``relax-vqa_amd``)
must fail loudly when ``librelax_hip.so`` is missing; it never falls back here.

Parity pinning status (see DESIGN.md "Oracle"):

* fragment path (absdiff, patch score, top-196 select, re-tile, merge):
  PINNED — bit-exact against the reference's own example PNG sets
  (``visualisation/visualisation_example/original_*``) and against the
  reference functions imported in the build container
  (``oracle/make_golden.py`` -> ``tests/golden/``).
* pooling (layer-stack GAP, pool stats, ViT token stats): PINNED — against the
  imported reference functions.
* ViT-B/16 forward: PINNED — against the reference's in-file
  ``VisionTransformer`` imported in the build container.
* ResNet-50 forward: the arithmetic lives in third-party torchvision==0.17.2
  (``requirements.txt:119``), absent from the reference checkout and from this
  image.  The restatement follows torchvision's published Bottleneck (v1.5)
  and is cross-checked against HuggingFace ``transformers`` ResNetModel (an
  independent implementation of the same architecture).  Against the reference
  itself: PARITY UNPINNED.
* Farneback optical flow + ``flow_to_rgb`` (``oracle/flow_ref.py``): OpenCV 4.9's algorithm restated (cv2 is absent from
  the image).  Pinned by TOLERANCE only - the reference's own ``*_residual_of*.png`` sets: >= 99.8 % of the flow image's
  bytes equal, >= 195 / 196 fragment positions (``tests/test_oracle_flow.py``, ``tests/test_gpu_reference_png_sets.py``).
"""
