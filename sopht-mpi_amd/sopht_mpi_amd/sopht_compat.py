"""The few names the reference imports from the un-vendored ``sopht`` package that its
callers touch directly (flow_simulators_mpi_3d.py:13-19): field index convention,
precision helpers and the pointwise operator generators, here backed by HIP kernels."""
import numpy as np

from ._lib import call, darray
from .device import as_ptr


class VectorField:
    """Component order of vector fields: x, y, z = 0, 1, 2 (flow_simulators_mpi_3d.py:154-157)."""

    @staticmethod
    def x_axis_idx():
        return 0

    @staticmethod
    def y_axis_idx():
        return 1

    @staticmethod
    def z_axis_idx():
        return 2


def get_real_t(precision="single"):
    if precision == "single":
        return np.float32
    if precision == "double":
        return np.float64
    raise ValueError("Precision argument must be single or double")


def get_test_tol(precision="single"):
    """Tolerance constant of ``sopht.utils.precision``; its value is not visible in the
    reference tree, so it is a module-level setting here (TEST_TOL)."""
    return TEST_TOL[precision]


TEST_TOL = {"single": 1e-4, "double": 1e-11}


def gen_elementwise_cross_product_pyst_kernel_3d(real_t):
    def elementwise_cross_product(result_field, field_1, field_2):
        call("smi_cross_product", result_field.ctx.handle, as_ptr(result_field), as_ptr(field_1), as_ptr(field_2))

    return elementwise_cross_product


def gen_set_fixed_val_pyst_kernel_3d(real_t, field_type="scalar"):
    if field_type == "scalar":

        def set_fixed_val(field, fixed_val):
            call("smi_set_fixed_val", field.ctx.handle, as_ptr(field), 1, darray([fixed_val]))

        return set_fixed_val

    def vector_field_set_fixed_val(vector_field, fixed_vals):
        call("smi_set_fixed_val", vector_field.ctx.handle, as_ptr(vector_field), 3, darray(fixed_vals))

    return vector_field_set_fixed_val


def gen_add_fixed_val_pyst_kernel_3d(real_t, field_type="scalar"):
    if field_type == "scalar":

        def add_fixed_val(sum_field, field, fixed_val):
            call("smi_add_fixed_val", field.ctx.handle, as_ptr(sum_field), as_ptr(field), 1, darray([fixed_val]))

        return add_fixed_val

    def vector_field_add_fixed_val(sum_field, vector_field, fixed_vals):
        call("smi_add_fixed_val", vector_field.ctx.handle, as_ptr(sum_field), as_ptr(vector_field), 3,
             darray(fixed_vals))

    return vector_field_add_fixed_val


def gen_elementwise_sum_pyst_kernel_3d(real_t):
    def elementwise_sum(sum_field, field_1, field_2):
        call("smi_elementwise_sum", sum_field.ctx.handle, as_ptr(sum_field), as_ptr(field_1), as_ptr(field_2), 1)

    return elementwise_sum
