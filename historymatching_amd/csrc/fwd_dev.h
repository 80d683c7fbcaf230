// fwd_dev.h -- device functions shared by the generic and the specialised forward kernels.
// Everything here is on the BIT-EXACT path (assembly of transmissibilities, fluxes): no FMA contraction,
// operations in the order of oracle/ressim.py.
#pragma once
#include "fwd.h"

template <typename T>
__device__ __forceinline__ void rel_perm(const FwdParams& p, T s, T& mw, T& mo) {
    // Listing RelPerm: S* = (s-swc)/(1-swc-sor); Mw = S*^2/vw; Mo = (1-S*)^2/vo
    if (p.fluid_default) {
        mw = s * s;
        T o = T(1) - s;
        mo = o * o;
    } else {
        T den = T((1.0 - p.swc) - p.sor);
        T S = (s - T(p.swc)) / den;
        mw = (S * S) / T(p.vw);
        T o = T(1) - S;
        mo = (o * o) / T(p.vo);
    }
}

// L = (Mt*K)**(-1) per cell, then harmonic-mean face transmissibilities TX (Nx+1,Ny) from the x-permeability, TY (Nx,Ny+1)
// from the y-permeability, zero on the boundary.  Kym == Km (the reference's case, set_perm HistoryMatch.py:164: Kx = Ky): one
// pass over L; otherwise L is rebuilt from Ky between the two face loops.  All threads of the workgroup must call this;
// contains barriers.
template <typename TS>
__device__ __forceinline__ void assemble_transmissibilities(const FwdParams& p, const TS* __restrict__ S,
                                                            const double* __restrict__ Km, const double* __restrict__ Kym,
                                                            double* __restrict__ L, double* __restrict__ TX, double* __restrict__ TY,
                                                            int tid, int T) {
    const int Nx = p.Nx, Ny = p.Ny, Nxy = p.Nxy;
    for (int j = tid; j < Nxy; j += T) {
        double mw, mo;
        rel_perm<double>(p, (double)S[j], mw, mo);
        double KM = (mw + mo) * Km[j];
        L[j] = 1.0 / KM;
    }
    __syncthreads();
    for (int f = tid; f < (Nx + 1) * Ny; f += T) {
        int ix = f / Ny, iy = f % Ny;
        TX[f] = (ix == 0 || ix == Nx) ? 0.0 : p.cx / (L[(ix - 1) * Ny + iy] + L[ix * Ny + iy]);
    }
    if (Kym != Km) {  // anisotropic: uniform over the workgroup
        __syncthreads();
        for (int j = tid; j < Nxy; j += T) {
            double mw, mo;
            rel_perm<double>(p, (double)S[j], mw, mo);
            double KM = (mw + mo) * Kym[j];
            L[j] = 1.0 / KM;
        }
        __syncthreads();
    }
    for (int f = tid; f < Nx * (Ny + 1); f += T) {
        int ix = f / (Ny + 1), iy = f % (Ny + 1);
        TY[f] = (iy == 0 || iy == Ny) ? 0.0 : p.cy / (L[ix * Ny + iy - 1] + L[ix * Ny + iy]);
    }
    __syncthreads();
}

// Vx = (P[i-1]-P[i])*TX, Vy = (P[:,j-1]-P[:,j])*TY on interior faces, 0 on the boundary.
__device__ __forceinline__ void face_fluxes(const FwdParams& p, const double* __restrict__ P, const double* __restrict__ TX,
                                            const double* __restrict__ TY, double* __restrict__ Vx, double* __restrict__ Vy,
                                            int tid, int T) {
    const int Nx = p.Nx, Ny = p.Ny;
    for (int f = tid; f < (Nx + 1) * Ny; f += T) {
        int ix = f / Ny, iy = f % Ny;
        Vx[f] = (ix == 0 || ix == Nx) ? 0.0 : (P[(ix - 1) * Ny + iy] - P[ix * Ny + iy]) * TX[f];
    }
    for (int f = tid; f < Nx * (Ny + 1); f += T) {
        int ix = f / (Ny + 1), iy = f % (Ny + 1);
        Vy[f] = (iy == 0 || iy == Ny) ? 0.0 : (P[ix * Ny + iy - 1] - P[ix * Ny + iy]) * TY[f];
    }
}
