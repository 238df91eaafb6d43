"""Host logic of the product (autograd Functions, fused blocks, models, samplers, operator pool) run on CPU tensors with
oracle-backed kernels (tests/cpu_kernels.py, test-only seam) and compared with the golden fixtures produced by the
reference.  The same checks run against the real HIP kernels in tests/test_blocks_gpu.py."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import product_checks as pc


@pytest.mark.parametrize("cname,C", pc.BLOCKS)
@pytest.mark.parametrize("opkind", ["pool", "coo2d", "coo3d"])
def test_blocks_match_reference(golden_dir, cpu_kernels, cname, C, opkind):
    if cname in ("AvgResNet2", "MlpResNet2") and opkind != "pool":
        pytest.skip("no sparse operator in this block")
    pc.check_block(golden_dir, cname, C, opkind, "cpu")


@pytest.mark.parametrize("tag", ["arap_dir", "arap_lap", "mnist_lap", "mnist_dir", "faust_lap"])
def test_models_match_reference(golden_dir, cpu_kernels, tag):
    pc.check_model(golden_dir, tag, "cpu")


def test_sparse_cat_functions_match_reference(golden_dir):
    pc.check_cat_functions(golden_dir)


def test_unfused_spmm_function_gradcheck(cpu_kernels):
    pc.check_spmm_autograd("cpu")


def test_arap_sampler_matches_padded_blockdiag(cpu_kernels):
    pc.check_arap_sampler("cpu")


def test_mnist_sampler_running_max(cpu_kernels):
    pc.check_mnist_sampler("cpu")


def test_pool_packed_assembly(golden_dir, cpu_kernels):
    pc.check_pool_packed(golden_dir, "cpu")


def test_inplace_edit_drops_the_activated_handoff(golden_dir, cpu_kernels):
    pc.check_inplace_edit_drops_handoff(golden_dir, "cpu")


@pytest.mark.parametrize("tag", ["arap_dir", "arap_lap", "mnist_lap", "mnist_dir", "faust_lap"])
def test_model_layers_match_reference_layer_by_layer(golden_dir, cpu_kernels, tag):
    pc.check_model_layers(golden_dir, tag, "cpu")


def test_model_variants_match_reference(golden_dir, cpu_kernels):
    pc.check_model_variants(golden_dir, "cpu")


def test_models_on_packed_batches(cpu_kernels):
    pc.check_packed_model("cpu")


def test_siamese_gradients_meet_once(cpu_kernels):
    pc.check_siamese_gradients_meet_once("cpu")


def test_mnist_driver_evaluation_mode(cpu_kernels):
    pc.check_mnist_evaluation_mode("cpu")


def test_forward_under_inference_mode(golden_dir, cpu_kernels):
    pc.check_inference_mode(golden_dir, "cpu")


def test_faust_amp_tower_from_files(golden_dir, cpu_kernels):
    pc.check_faust_amp_tower(golden_dir, "cpu")


def test_graph_signature_carries_batch_constants():
    """PairBatch.NA / NB are baked into the captured kernels' arguments: they are part of the signature GraphedStep compares."""
    from surfacenetworks_amd import graphs

    class B:
        def __init__(self, n, consts):
            self.t, self.c = [torch.zeros(n)], consts

        def graph_tensors(self):
            return self.t

        def graph_constants(self):
            return self.c

    assert graphs.batch_signature(B(3, (5, 6))) == graphs.batch_signature(B(3, (5, 6)))
    assert graphs.batch_signature(B(3, (5, 6))) != graphs.batch_signature(B(3, (5, 7)))
    assert graphs.batch_signature_constants(B(3, (5, 7))) == ("const", 5, 7)
