"""Import stand-in for scikit-image, which is absent from this image.  TEST INFRASTRUCTURE ONLY.
The reference's loss/hausdorff.py and loss/edge_loss.py import `skimage.filters.sobel` at module level; the functions
the fixtures are generated from (get_edg_c, edge_conv_loss, get_hot_enc) never call it."""
