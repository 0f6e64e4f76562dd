#pragma once
// threadIdx/blockIdx are built in under hipcc
