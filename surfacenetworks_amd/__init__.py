"""surfacenetworks_amd — MI355X-native Surface-Network operator layer (Lap/Dirac ResNet block hot path).

Public surface:
    surfacenetworks_amd.utils_pt     drop-in for the reference's `utils.utils_pt`
    surfacenetworks_amd.functional   spmm / fused block stages (torch.autograd.Function over the C-ABI)
    surfacenetworks_amd.operators    SparseOperator, OperatorPool (device-resident CSR / CSR^T / BSR4)
    surfacenetworks_amd.mesh_ops     sparse-direct Laplacian / Dirac construction + synthetic meshes
    surfacenetworks_amd.arap / mesh_mnist / dense_correspondence   the three task harnesses
    surfacenetworks_amd.dp           mesh-batch sharding + flat-bucket RCCL gradient all-reduce
"""
__version__ = "0.1.0"
