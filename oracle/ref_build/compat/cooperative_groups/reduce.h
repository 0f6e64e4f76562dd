#pragma once
// cg::reduce is included but never used by the reference kernels
