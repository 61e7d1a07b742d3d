"""Configuration schema for the MI355X inference path.

Drop-in for the reference's ``parameters.py`` (reference: parameters.py:12-45 field
list, :52-78 PANOPTIC preset, :79-123 ARPLAB preset).  Every hot-path module of the
reference does ``from parameters import parameters`` and reads the fields below by
name, so the field names, their meaning and the preset values are kept; the
construction is our own (a preset table + ``select``) so that tests can switch
presets and build N-camera stress variants without editing a module constant.

The numeric values are calibration data of the CMU-Panoptic / ARP-lab rigs.
"""
from collections import namedtuple

FORMAT = 'COCO'                       # 18 joints = COCO-17 + neck (reference parameters.py:3-8)
_N_JOINTS = {'COCO': 18, 'BODY_25': 25}
JOINT_LIST = list(range(_N_JOINTS[FORMAT]))

_FIELDS = (
    'image_width', 'image_height', 'cameras', 'camera_names', 'widths', 'heights',
    'fx', 'fy', 'cx', 'cy', 'r_s', 'r_w', 'c_s', 'c_w', 'kd0', 'kd1', 'kd2', 'p1', 'p2',
    'joint_list', 'numbers_per_joint', 'numbers_per_joint_for_loss', 'transformations_path',
    'used_cameras', 'used_cameras_skeleton_matching', 'used_joints', 'min_number_of_views',
    'format', 'graph_alternative', 'axes_3D',
)
TrackerParameters = namedtuple('TrackerParameters', _FIELDS, defaults=(None,) * len(_FIELDS))

_USED_JOINTS = [0] + list(range(5, 18))

_PANOPTIC_NAMES = ['tracker' + s for s in 'abcde']


def _panoptic():
    return TrackerParameters(
        image_width=1920, image_height=1080,
        cameras=list(range(5)), camera_names=list(_PANOPTIC_NAMES),
        fx=[1395.59, 1395.94, 1395.31, 1591.32, 1572.31],
        fy=[1392.03, 1392.22, 1391.77, 1587.2, 1567.51],
        cx=[950.046, 950.459, 966.65, 940.617, 942.938],
        cy=[564.906, 547.877, 562.988, 560.913, 559.888],
        kd0=[-0.28619, -0.279874, -0.284888, -0.232872, -0.237061],
        kd1=[0.179547, 0.166215, 0.179936, 0.194125, 0.18403],
        kd2=[-0.0451919, -0.035049, -0.0468637, 0.0125375, 0.0149481],
        p1=[-0.00010526, -0.000189415, -0.000119731, 4.22e-05, -0.000448556],
        p2=[6.45495e-05, 0.00107791, 0.000701704, 0.000877748, 0.00062731],
        joint_list=JOINT_LIST, numbers_per_joint=14, numbers_per_joint_for_loss=4,
        transformations_path='../tm_panoptic.pickle',
        used_cameras=list(_PANOPTIC_NAMES),
        used_cameras_skeleton_matching=list(_PANOPTIC_NAMES),
        used_joints=list(_USED_JOINTS), min_number_of_views=2, format=FORMAT,
        graph_alternative='3',
        # (coordinate index, axis direction) per drawing axis; 'Y' picks the vertical
        # coordinate used by the triangulation median filter.
        axes_3D={'X': (0, 1.), 'Y': (2, 1.), 'Z': (1, -1.)},
    )


def _arplab():
    f = 848. / 1280.
    zf = 720. / 1080.
    names = ['trackera', 'trackerb', 'trackerc', 'trackerd', 'orinbot_l', 'orinbot_r']
    zfx, zcx, zcy = 1097.2998046875 * zf, 953.3253173828125 * zf, 553.707763671875 * zf
    six = [0] * 6
    return TrackerParameters(
        image_width=1280, image_height=720,
        cameras=list(range(6)), camera_names=list(names),
        kd0=[0.] * 7, kd1=[0.] * 7, kd2=[0.] * 7, p1=[0.] * 7, p2=[0.] * 7,
        widths=[1280] * 6, heights=[720] * 6,
        fx=[634.0370 * f, 633.6757 * f, 636.5411 * f, 635.4050 * f, zfx, zfx],
        fy=[633.5662 * f, 633.0649 * f, 636.1349 * f, 634.5941 * f, zfx, zfx],
        cx=[631.7626 * f, 635.7685 * f, 638.4467 * f, 638.3454 * f, zcx, zcx],
        cy=[355.3067 * f, 358.7285 * f, 370.3130 * f, 362.9503 * f, zcy, zcy],
        r_s=list(six), r_w=[720] * 6, c_s=list(six), c_w=[1280] * 6,
        joint_list=JOINT_LIST, numbers_per_joint=14, numbers_per_joint_for_loss=4,
        transformations_path='../tm_arp.pickle',
        used_cameras=list(names), used_cameras_skeleton_matching=list(names),
        used_joints=list(_USED_JOINTS), min_number_of_views=2, format=FORMAT,
        graph_alternative='3',
        axes_3D={'X': (0, 1.), 'Y': (1, 1.), 'Z': (2, -1.)},
    )


def ring_variant(n_cameras=23):
    """Stress preset of BASELINE.json config 5: `n_cameras` cameras on a ring, intrinsics and
    lens coefficients cycled from the five Panoptic cameras (not a reference preset; the
    extrinsics come from synthetic.ring_transform_manager)."""
    base = _panoptic()
    names = ['ring%02d' % i for i in range(n_cameras)]

    def cyc(v):
        return [v[i % len(v)] for i in range(n_cameras)]
    return base._replace(
        cameras=list(range(n_cameras)), camera_names=list(names),
        fx=cyc(base.fx), fy=cyc(base.fy), cx=cyc(base.cx), cy=cyc(base.cy),
        kd0=cyc(base.kd0), kd1=cyc(base.kd1), kd2=cyc(base.kd2), p1=cyc(base.p1), p2=cyc(base.p2),
        transformations_path='../tm_ring%d.pickle' % n_cameras,
        used_cameras=list(names), used_cameras_skeleton_matching=list(names))


def _arplab_robot():
    """ARPLAB with the reference's "models using only the robot cameras" lines active
    (reference parameters.py:110-112): six cameras configured, two used."""
    robot = ['orinbot_l', 'orinbot_r']
    return _arplab()._replace(used_cameras=list(robot), used_cameras_skeleton_matching=list(robot))


PRESETS = {'PANOPTIC': _panoptic, 'ARPLAB': _arplab, 'ARPLAB_ROBOT': _arplab_robot, 'RING23': lambda: ring_variant(23)}

CONFIGURATION = 'PANOPTIC'            # values = {PANOPTIC, ARPLAB} (reference parameters.py:47)


def select(name):
    """Return the preset `name` (does not change the module-level `parameters`)."""
    if name not in PRESETS:
        raise KeyError('NO VALID CONFIGURATION: %r' % (name,))
    p = PRESETS[name]()
    assert len(p.cameras) == len(p.camera_names), \
        "The number of cameras must be equal in 'cameras' and 'camera_names'"
    return p


parameters = select(CONFIGURATION)
