"""`isaacgym.gymutil`: the helpers the reference's task files call (quadcopter.py:201 / ingenuity.py:230 `_indent_xml` on the MJCF they
generate; ball_balance.py:433-456 the debug-visualisation geometry, only with enableDebugVis)."""


def parse_arguments(*args, **kwargs):
    raise NotImplementedError("gymutil.parse_arguments: the reference's train.py uses Hydra; not part of the shim")


def _indent_xml(elem, level=0):
    """pretty-print an ElementTree in place (whitespace only)"""
    i = "\n" + level * "  "
    if len(elem):
        if not elem.text or not elem.text.strip():
            elem.text = i + "  "
        if not elem.tail or not elem.tail.strip():
            elem.tail = i
        for elem in elem:
            _indent_xml(elem, level + 1)
        if not elem.tail or not elem.tail.strip():
            elem.tail = i
    elif level and (not elem.tail or not elem.tail.strip()):
        elem.tail = i


class AxesGeometry:          # debug visualisation only (headless engine: nothing is drawn)
    def __init__(self, scale=1.0, pose=None):
        self.scale, self.pose = scale, pose


class WireframeSphereGeometry:
    def __init__(self, radius=1.0, num_lats=8, num_lons=8, pose=None, color=None, color2=None):
        self.radius = radius


def draw_lines(geom, gym, viewer, env, pose):
    pass
