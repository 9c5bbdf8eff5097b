s/extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw\[\];/unsigned char* smem_raw = cpuhip_dyn_lds;/
s/const h8 vf = \*reinterpret_cast<const h8\*>(sVT/cpuhip::ctx.wave_bar->arrive_and_wait(); const h8 vf = *reinterpret_cast<const h8*>(sVT/
