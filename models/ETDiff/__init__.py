"""Import paths of the reference (`models.ETDiff.*`), served by the MI355X-native package timediff_amd."""
