"""NumPy restatement of the reference's physical models, one particle at a time, written the way the reference writes them.

TEST INFRASTRUCTURE ONLY (tests/).  The product defines its batched models in pgas_amd/experiments.py against an array
namespace; this file restates the same functions INDEPENDENTLY from the reference sources, per particle and with the
reference's own expression structure, so that the model arithmetic of the marginalised family (RK4 steps, tyre model,
friction model, output maps) is checked against something that is not the product's own code.
PARITY UNPINNED against JAX like every other oracle here: the reference ships no fixture for these
functions; what this pins is "the product's models == a literal reading of the reference's formulas".

Paths are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np


# ---------------------------------------------------------------------------------------------- src/SingleMassOscillator.py
class SMO:
    m, c1, c2, d1, d2 = 0.2, 5.0, 2.0, 0.4, 0.4      # :17-21
    dt = 0.02                                        # :78

    @staticmethod
    def F_spring(x):                                 # :24-25
        return SMO.c1 * x + SMO.c2 * x**3

    @staticmethod
    def F_damper(dx):                                # :28-29
        return SMO.d1 * dx * (1 / (1 + SMO.d2 * dx * np.tanh(dx)))

    @staticmethod
    def dx(x, F, F_sd, m=None):                      # :32-33
        m = SMO.m if m is None else m
        return np.hstack([x[1], (-F_sd + F) / m])

    @staticmethod
    def f_x(x, F, F_sd, dt):                         # :36-44
        k1 = SMO.dx(x, F, F_sd)
        k2 = SMO.dx(x + dt / 2.0 * k1, F, F_sd)
        k3 = SMO.dx(x + dt / 2.0 * k2, F, F_sd)
        k4 = SMO.dx(x + dt * k3, F, F_sd)
        return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    @staticmethod
    def f_y(x):                                      # :47-48
        return x[0]

    # StateSpaceModel wiring, :101-109
    @staticmethod
    def transition_model(state, input, *int_var):
        return SMO.f_x(state, input, int_var[0], SMO.dt)

    @staticmethod
    def output_model(state, input, *int_var):
        return SMO.f_y(state)


# ---------------------------------------------------------------------------------------------- src/Vehicle.py
class Vehicle:
    m, I_zz, l_f, l_r, g, mu_x = 1720.0, 1827.5, 1.16, 1.47, 9.81, 0.9   # :17-22
    mu, B, C, E = 0.9, 10.0, 1.9, 0.97                                    # :23-26
    dt = 0.02                                                             # :183

    @staticmethod
    def f_Fz(m, l_f, l_r, g):                        # :30-36
        l_total = l_f + l_r
        mg = m * g
        return mg * l_r / l_total, mg * l_f / l_total

    @staticmethod
    def mu_y(alpha):                                 # :40-47
        V = Vehicle
        return V.mu * np.sin(V.C * np.arctan(V.B * (1 - V.E) * np.tan(alpha) + V.E * np.arctan(V.B * np.tan(alpha))))

    @staticmethod
    def f_alpha(x, u):                               # :51-58
        V = Vehicle
        vx_f = u[1]
        vy_f = x[1] + x[0] * V.l_f
        vx_r = u[1]
        vy_r = x[1] - x[0] * V.l_r
        return u[0] - np.arctan(vy_f / vx_f), -np.arctan(vy_r / vx_r)

    @staticmethod
    def dx(x, u, mu_yf, mu_yr):                      # :62-86
        V = Vehicle
        F_zf, F_zr = V.f_Fz(V.m, V.l_f, V.l_r, V.g)
        dv_y = 1 / V.m * (F_zf * mu_yf * np.cos(u[0]) + F_zr * mu_yr + F_zf * V.mu_x * np.sin(u[0])) - u[1] * x[0]
        ddpsi = 1 / V.I_zz * (V.l_f * F_zf * mu_yf * np.cos(u[0]) - V.l_r * F_zr * mu_yr + V.l_f * F_zf * V.mu_x * np.sin(u[0]))
        return np.hstack([ddpsi, dv_y])

    @staticmethod
    def f_x(x, u, mu_yf, mu_yr, dt):                 # :90-100
        V = Vehicle
        k1 = V.dx(x, u, mu_yf, mu_yr)
        k2 = V.dx(x + dt * k1 / 2.0, u, mu_yf, mu_yr)
        k3 = V.dx(x + dt * k2 / 2.0, u, mu_yf, mu_yr)
        k4 = V.dx(x + dt * k3, u, mu_yf, mu_yr)
        return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    @staticmethod
    def f_y(x, u, mu_yf, mu_yr):                     # :104-131
        V = Vehicle
        F_zf, F_zr = V.f_Fz(V.m, V.l_f, V.l_r, V.g)
        dv_y = 1 / V.m * (F_zf * mu_yf * np.cos(u[0]) + F_zr * mu_yr + F_zf * V.mu_x * np.sin(u[0])) - u[1] * x[0]
        return np.tanh(np.hstack([x[0], dv_y]))

    # StateSpaceModel wiring, :211-221
    @staticmethod
    def transition_model(state, input, *int_var):
        return Vehicle.f_x(state, input, int_var[0], int_var[1], Vehicle.dt)

    @staticmethod
    def output_model(state, input, *int_var):
        return Vehicle.f_y(state, input, int_var[0], int_var[1])


# ---------------------------------------------------------------------------------------------- src/EMPS.py
class EMPS:
    M = 95.11                                        # :156
    dt = 0.01                                        # 1 kHz data decimated x10, :59-65

    @staticmethod
    def dx(x, tau, F):                               # :160-165
        dq = x[1]
        ddq = (tau - F) / EMPS.M
        return np.hstack([dq, ddq])

    @staticmethod
    def dx_linModel(x, tau):                         # :168-172
        dq = x[1]
        ddq = (tau - 203.5 * x[1] - 20.39 * np.sign(x[1]) + 3.16) / 95.11
        return np.hstack([dq, ddq])

    @staticmethod
    def f_x(x, tau, F, dt=None):                     # :176-182
        dt = EMPS.dt if dt is None else dt
        k1 = EMPS.dx(x, tau, F)
        k2 = EMPS.dx(x + dt * k1 / 2, tau, F)
        k3 = EMPS.dx(x + dt * k2 / 2, tau, F)
        k4 = EMPS.dx(x + dt * k3, tau, F)
        return x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

    @staticmethod
    def f_x_linModel(x, tau, dt):                    # :185-192
        k1 = EMPS.dx_linModel(x, tau)
        k2 = EMPS.dx_linModel(x + dt * k1 / 2, tau)
        k3 = EMPS.dx_linModel(x + dt * k2 / 2, tau)
        k4 = EMPS.dx_linModel(x + dt * k3, tau)
        return x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

    @staticmethod
    def f_y(x):                                      # :196-197
        return x[0]

    # StateSpaceModel wiring, :201-209
    @staticmethod
    def transition_model(state, input, *int_var):
        return EMPS.f_x(state, input, int_var[0], EMPS.dt)

    @staticmethod
    def output_model(state, input, *int_var):
        return EMPS.f_y(state)


# ---------------------------------------------------------------------------------------------- src/Toy_Example.py
class Toy:
    @staticmethod
    def f_x(x):                                      # :18-19   (jnp.sinc(x) = sin(pi x) / (pi x))
        return 10 * np.sinc(x / 7)

    @staticmethod
    def transition_model(state, input, *int_var):    # :69
        return int_var[0]

    @staticmethod
    def output_model(state, input, *int_var):        # :70 with f_y = identity (:22-23)
        return int_var[0]
