/* placeholder, filled in below */
