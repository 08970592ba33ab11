"""`Track` — the per-feature observation record of the reference (`/root/reference/track.py:1-41`).

Same constructor, methods and dict-ordering behaviour, so objects are interchangeable with the reference's:
observations live in an insertion-ordered ``{frame_ID: (x, y)}``; re-writing an existing frame keeps its
position; `getTriangulationData` returns the first and last inserted frames.  The bulk pipeline
(`meatmodeler_amd.pipeline`) keeps the same information as CSR index tensors instead and only materialises
`Track` objects on request.
"""


class Track:
    __slots__ = ("coordinates", "point", "updated")

    def __init__(self, prev_frame_ID, feature, frame_ID, correspondent):
        self.coordinates = {}
        self.coordinates[prev_frame_ID] = feature
        self.coordinates[frame_ID] = correspondent
        self.point = None
        self.updated = False

    # -- state flag toggled by pointTracking (processor.py:221,234-235) --
    def update(self, frame_ID, correspondent):
        self.coordinates[frame_ID] = correspondent
        self.updated = True

    def reset(self):
        self.updated = False

    def wasUpdated(self):
        return self.updated

    # -- accessors --
    def getCoordinate(self, frame_ID):
        return self.coordinates.get(frame_ID)

    def getCoordinates(self):
        return self.coordinates

    def getTriangulationData(self):
        first = next(iter(self.coordinates))
        last = next(reversed(self.coordinates))
        return first, last, self.coordinates[first], self.coordinates[last]

    def setPoint(self, point):
        self.point = point

    def getPoint(self):
        return self.point
