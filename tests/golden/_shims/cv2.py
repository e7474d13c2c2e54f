"""Stand-in for OpenCV: team_code_v2/lav_agent_fast.py only draws its debug video with it (visualize, :459-517).
The fixture generator replaces visualize(); the one name evaluated at class-definition time is the font constant."""
FONT_HERSHEY_SIMPLEX = 0
COLOR_GRAY2RGB = 8

# --- the three calls of lav/utils/datasets (imdecode, getRotationMatrix2D, warpAffine): delegated to lav_amd.data.image, this
# repository's restatement of them - so a reference loader run over this stand-in pins the loader LOGIC, not OpenCV.
IMREAD_GRAYSCALE, IMREAD_COLOR, INTER_LINEAR = 0, 1, 1


def imdecode(buf, mode):
    from lav_amd.data import image
    return image.imdecode(buf, mode)


def getRotationMatrix2D(center, angle, scale):
    from lav_amd.data import image
    return image.rotation_matrix_2d(center, angle, scale)


def warpAffine(img, M, dsize, flags=INTER_LINEAR):
    from lav_amd.data import image
    assert flags == INTER_LINEAR and tuple(dsize) == tuple(img.shape[1::-1])
    return image.warp_affine_linear(img, M)
