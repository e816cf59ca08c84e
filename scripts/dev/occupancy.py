"""Resident workgroups per CU of the main kernels at their launch shapes (mi_icp_debug_occupancy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cupoch_amd import _lib
L = _lib.load()
torch.cuda.init()
names = ["kd_build_groups (1024 threads)", "nn_packet_kernel<seeded> (64)", "nn_packet_kernel<root> (64)", "reduce_pt2pl_kernel<4,1> (256)",
         "leaf_halo_build (64)", "rs_scatter_pay<8> (512)", "voxel_means_wave (64)"]
for i, n in enumerate(names):
    print("occupancy: %-40s %d workgroups per CU" % (n, L.mi_icp_debug_occupancy(i)))
