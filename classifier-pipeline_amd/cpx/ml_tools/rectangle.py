"""Axis-aligned rectangle with the reference's geometry API
(reference src/ml_tools/rectangle.py:6-177): left/top setters keep the
right/bottom edge, crop() clamps edge by edge, enlarge() grows then crops."""


class Rectangle:
    __slots__ = ("x", "y", "width", "height")

    def __init__(self, x, y, width, height):
        self.x, self.y, self.width, self.height = x, y, width, height

    # ---- constructors / conversions ----
    @classmethod
    def from_ltrb(cls, left, top, right, bottom):
        return cls(left, top, right - left, bottom - top)

    def to_ltrb(self):
        return [self.left, self.top, self.right, self.bottom]

    def to_ltwh(self):
        return [self.left, self.top, self.width, self.height]

    def copy(self):
        return Rectangle(self.x, self.y, self.width, self.height)

    # ---- edges ----
    @property
    def left(self):
        return self.x

    @left.setter
    def left(self, value):
        right = self.right
        self.x = value
        self.width = right - value

    @property
    def top(self):
        return self.y

    @top.setter
    def top(self, value):
        bottom = self.bottom
        self.y = value
        self.height = bottom - value

    @property
    def right(self):
        return self.x + self.width

    @right.setter
    def right(self, value):
        self.width = value - self.x

    @property
    def bottom(self):
        return self.y + self.height

    @bottom.setter
    def bottom(self, value):
        self.height = value - self.y

    @property
    def mid_x(self):
        return self.x + self.width / 2

    @property
    def mid_y(self):
        return self.y + self.height / 2

    @property
    def mid(self):
        return (self.mid_x, self.mid_y)

    @property
    def area(self):
        return int(self.width) * self.height

    @property
    def elongation(self):
        return max(self.width, self.height) / min(self.width, self.height)

    # ---- operations ----
    def overlap_area(self, other):
        xo = max(0, min(self.right, other.right) - max(self.left, other.left))
        yo = max(0, min(self.bottom, other.bottom) - max(self.top, other.top))
        return xo * yo

    def crop(self, bounds):
        self.left = min(bounds.right, max(self.left, bounds.left))
        self.top = min(bounds.bottom, max(self.top, bounds.top))
        self.right = max(bounds.left, min(self.right, bounds.right))
        self.bottom = max(bounds.top, min(self.bottom, bounds.bottom))

    def enlarge(self, border, max=None):
        self.left -= border
        self.right += border
        self.top -= border
        self.bottom += border
        if max:
            self.crop(max)

    def subimage(self, image):
        return image[self.top : self.top + self.height, self.left : self.left + self.width]

    def contains(self, x, y):
        # (sic) the reference's vertical test is inverted; kept for drop-in behaviour
        return self.left <= x and self.right >= x and self.top >= y and self.bottom <= y

    def __repr__(self):
        return "(x{0},y{1},x2{2},y2{3})".format(self.left, self.top, self.right, self.bottom)

    def __str__(self):
        return "<(x{0},y{1})-h{2}xw{3}>".format(self.x, self.y, self.height, self.width)

    def meta_dictionary(self):
        return {"x": self.x, "y": self.y, "width": self.width, "height": self.height}
