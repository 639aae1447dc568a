"""Kernel-point dispositions for KPConv.

Mirror of the *loading* half of the reference's load_kernels()
(/root/reference/src/models/backbone_kpconv/kernels/kernel_points.py:387-469): read the optimised
unit-radius disposition, apply a random z-rotation and N(0, 0.01) jitter from the NumPy global RNG,
scale by the convolution radius.  The optimisation half (Lloyd / gradient descent, :65-383) is out
of scope: the only disposition either shipped config uses (15 points, 'center', 3-D) is embedded
below as data (values of src/kernels/dispositions/k_015_center_3D.ply, float64).
At inference the kernel points come from the checkpoint (they are a non-trainable Parameter,
kpconv_blocks.py:266), so this only matters for random-init construction.
"""
import numpy as np

K015_CENTER = np.array([
    (0.0, 0.0, 0.0),
    (0.36145941026597067, 0.48239212397030184, -0.27125260188676675),
    (-0.48163767397727025, -0.22902570148057874, 0.3905206150186844),
    (0.4372927010555656, -0.49427528002130244, 0.03741766626651985),
    (-0.5198507833083387, -0.2979149446938551, -0.279168209843934),
    (-0.12344802259344391, -0.6311186121210636, 0.15291476004445942),
    (-0.6096160563430828, 0.24541086682383462, -0.07123770770511689),
    (-0.1689108743398532, 0.635534777653165, -0.06707186148862122),
    (0.6399171252503684, 0.04467887639897198, 0.1595112613255694),
    (-0.17213346075329766, 0.20048248013725176, -0.5981755025792743),
    (0.2818393803824886, 0.44664145088924706, 0.39750599834002404),
    (0.010631422808218766, -0.45118795265919565, -0.48296001481067796),
    (0.17213346082322145, -0.20048248011358744, 0.598175502567091),
    (0.4512246987754341, -0.06093370553907477, -0.47918305004457534),
    (-0.278901328009585, 0.3097981007681546, 0.5130031447903965),
], dtype=np.float64)


def load_kernels(radius, num_kpoints, dimension=3, fixed='center'):
    if (num_kpoints, dimension, fixed) != (15, 3, 'center'):
        raise NotImplementedError(
            'only the 15-point "center" 3-D disposition is embedded (the one both reference configs use)')
    kernel_points = K015_CENTER.copy()
    theta = np.random.rand() * 2 * np.pi                      # kernel_points.py:434
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)   # :441-443
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)   # :461
    kernel_points = radius * kernel_points                    # :464
    kernel_points = np.matmul(kernel_points, R)               # :467
    return kernel_points.astype(np.float32)
