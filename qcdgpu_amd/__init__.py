"""qcdgpu_amd -- MI355X-native SU(N) lattice-gauge sweep engine (heat-bath update, plaquette,
action, Polyakov loop) behind a C-ABI; see include/qcdgpu_amd.h."""
from .capi import (Context, Ring, RING_COPY, RING_RCCL, OnContext, ON_DEFAULTS, QcdError, START_COLD, START_GID, START_HOT, LIB_PATH,  # noqa: F401
                   host_gid_link, host_gid_randoms, host_ranlux_stream, prng_check, math_check, trig_check, rng_share_for)
