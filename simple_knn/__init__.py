"""Drop-in `simple_knn` package: `from simple_knn._C import dist3knn, dist10knn, meanDistFromReferencePcd`
[REF /root/reference/scene/gaussian_model.py:16; scene/mask_gaussian.py:21;
inpainting_pipeline/2_condition_preparation/2_generate_inpainted_mask.py:27] on the HIP kernels of streetunveiler_amd/csrc/knn.hip."""
