"""`north_star`: "the existing mesh_mnist / as_rigid_as_possible / dense_correspondence training scripts run unmodified apart
from an import swap".  Here the REFERENCE's own, unmodified model files (src/as_rigid_as_possible/models.py:18,108-152,
src/mesh_mnist/models.py:18,122-159, src/dense_correspondence/models.py:18,184-203) are loaded with
`utils.utils_pt` resolving to surfacenetworks_amd.utils_pt, fed what their drivers feed them (torch sparse COO operators from
sparse_diag_cat / sparse_cat), run forward + backward on the golden batch (oracle-backed kernels: the CPU seam), and held to
the same criterion as the product's own model classes: as close to the float64 oracle as the reference's stored fp32 run.

The reference checkout never travels to the GPU box: skipped when /root/reference is absent."""
import os

import pytest

import product_checks as pc

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference checkout not present (it never travels)")


@pytest.mark.parametrize("tag,opkind", [("arap_dir", "coo2d"), ("arap_lap", "coo2d"), ("mnist_lap", "coo2d"),
                                        ("mnist_dir", "coo3d"), ("faust_lap", "coo2d")])
def test_reference_model_files_run_on_the_swapped_operator_layer(golden_dir, reference_models, tag, opkind):
    """mnist_dir takes the 3-D `sparse_cat` operators its driver builds (mesh_mnist/main.py:109-111; the class reads
    DiA.size(2), models.py:142); the others the 2-D block-diagonal ones."""
    r = reference_models
    classes = {"arap_dir": r["as_rigid_as_possible"].DirModel, "arap_lap": lambda: r["as_rigid_as_possible"].Model(15),
               "mnist_lap": r["mesh_mnist"].Model, "mnist_dir": r["mesh_mnist"].DirModel,
               "faust_lap": lambda: r["dense_correspondence"].SiameseModel("lap", 15)}
    pc.check_model(golden_dir, tag, "cpu", opkind=opkind, classes=classes)


def test_swapped_models_are_built_from_the_product_blocks(reference_models):
    import surfacenetworks_amd.utils_pt as U

    m = reference_models["as_rigid_as_possible"].DirModel()
    assert isinstance(m.rn0, U.DirResNet2) and isinstance(m.rn1, U.AvgResNet2) and isinstance(m.conv1, U.GraphConv1x1)
    m = reference_models["dense_correspondence"].SiameseModel("lap", 15)
    assert isinstance(m.model.rn0, U.LapResNet2)
